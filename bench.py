#!/usr/bin/env python3
"""HSI cubes/sec, forward+backward, for the HyperPRI CubeNET-64 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] / configs[3]): CubeNET(238, 1, first_depth=64, bilinear=False) in train
mode on synthetic 608x968x238 fp32 cubes, per-rank batch 2 (params_HyperPRI.py:178), loss =
BCEWithLogitsLoss (params_HyperPRI.py:223), one step = forward + loss + backward (+ gradient all-reduce
for N > 1; optimizer excluded).  Inputs are generated on the device by the counter-based
generator before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, W, BANDS, BATCH = 608, 968, 238, 2
GFLOP_PER_CUBE = 2910.17          # fwd 1023.84 + bwd 1886.32 (measured on the reference modules)
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 matrix peak (v_mfma_f32_32x32x2_f32)


def synth_init_(net):
    """Generator-defined weights on the device: k-th parameter = (2u(1000+k)-1)/sqrt(fan_in); BN gamma=1, beta=0."""
    from hyperpri_amd.engine import synth_fill_
    bn_params = set()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            bn_params.add(id(m.weight)); bn_params.add(id(m.bias))
    fan_in = 1
    with torch.no_grad():
        for k, p in enumerate(net.parameters()):
            if id(p) in bn_params:
                continue                      # default init already gamma=1, beta=0
            if p.dim() >= 2:
                fan_in = p.shape[1] * int(math.prod(p.shape[2:]))
            synth_fill_(p.data, 1000 + k, mode=2, scale=1.0 / math.sqrt(fan_in))


def _lib_stamp():
    """Source stamp of the loaded HIP library (hyperpri_amd/build.py writes it next to the .so)."""
    try:
        from hyperpri_amd import build as _B
        return open(_B.LIB + ".stamp").read().strip()
    except Exception:
        return None


def _replayable(path):
    """(kernels, note): the per-kernel figures of a committed PMC file, or (None, why not) when the file was recorded on another
    build of the library (replayed counters must not outlive the kernels they were read from)."""
    try:
        doc = json.load(open(path))
    except Exception:
        return None, "unreadable"
    have, want = doc.get("library_stamp"), _lib_stamp()
    if have is None or want is None or have != want:
        return None, (f"{os.path.relpath(path, ROOT)} was recorded on another build of the library (stamp {str(have)[:12]} vs loaded "
                      f"{str(want)[:12]}): not replayed; re-run tools/measure.sh + tools/pmc_traffic.py")
    return doc["kernels"], None


def pmc_traffic(tag):
    """(HBM bytes per launch of the dominant kernel, source file) REPLAYED from the committed PMC passes
    (profiles/*_pmc_traffic.json, produced by tools/measure.sh + tools/pmc_traffic.py with the gfx950 FETCH_SIZE
    correction): hardware counters cannot be read from inside this process, so the figure is not measured in this run
    and the JSON line says which file it came from.  (None, None) if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))   # the fp32 bench passes only
    if not files:
        return None, None
    kern, why = _replayable(files[-1])
    if kern is None:
        return None, "NOT REPLAYED: " + why
    src = os.path.relpath(files[-1], ROOT)
    if tag.startswith("conv_fwd<3,2x2"):
        key = "conv_fwd_kernel<3, 2, 2, 0, 0>"
    elif tag.startswith("conv_fwd<3,4x1"):
        key = "conv_fwd_kernel<3, 4, 1, 0, 0>"
    elif tag.startswith("conv_wgrad<3"):
        key = "conv_wgrad_kernel<3, 1, 1, 0>"
    elif tag.startswith("conv_winograd_f32"):
        key = "conv_wino4_kernel" if any("conv_wino4_kernel" in k for k in kern) else "conv_wino_kernel"
    elif tag.startswith("conv_wgrad_winograd_f32"):
        key = "conv_wino_wgrad_kernel"
    else:
        return None, src
    for k, v in kern.items():
        if key in k:
            return round(v["hbm_bytes_per_launch"]), src
    return None, src


def first_conv_traffic():
    """(measured HBM bytes per launch of the bf16-plane 238->64 convolution, source file), REPLAYED from the newest committed
    profiles/rNN_first_conv_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/first_conv.py, gfx950
    correction applied by tools/pmc_traffic.py); (None, None) if no such file is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_first_conv_pmc_traffic.json")))
    if not files:
        return None, None
    kern, why = _replayable(files[-1])
    if kern is None:
        return None, "NOT REPLAYED: " + why
    src = os.path.relpath(files[-1], ROOT) + " (replayed: rocprofv3 --pmc passes of `tools/first_conv.py 10 bf16_out_only`, not measured in this run)"
    for k, v in kern.items():
        if "conv_bf16v3" in k or (("conv_bf16v2" in k) and not any("conv_bf16v3" in q for q in kern)):
            return round(v["hbm_bytes_per_launch"]), src
    return None, src


def host_cores():
    """Cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (the GPU box
    exposes all host CPUs in the affinity mask but grants a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(math.ceil(q / per))))
            break
        except Exception:
            continue
    env = os.environ.get("HPRI_CPU_THREADS")
    if env:
        n = int(env)
    return min(n, 16)   # the 1-GPU box grants a 16-core share whatever the affinity mask says


def cpu_baseline():
    """The CPU oracle (validated against the reference modules) on this box's host cores: CubeNET-64,
    batch 1, one warm-up + three timed forward+backward steps (about 30 s of CPU work)."""
    from collections import OrderedDict
    import numpy as np
    from oracle import hyperpri_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = O.synth_state_dict(O.cubenet_shapes(BANDS, 1, 64))
    x = torch.from_numpy(O._u(1234, BANDS * H * W).reshape(1, 1, BANDS, H, W).copy())
    mask = (torch.from_numpy(O._u(4321, H * W).reshape(1, 1, H, W).copy()) > 0.9).float()
    times = []
    for _ in range(4):
        work = OrderedDict((k, v.clone()) for k, v in sd.items())
        t0 = time.perf_counter()
        O.train_step(O.cubenet_forward, work, x, mask, first_depth=64)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / len(times[1:])
    return {"value": 1.0 / t, "unit": "cubes/s", "cores": cores, "kind": "port",
            "sample": "1 warm-up + 3 timed fwd+bwd steps of CubeNET-64 on ONE 238x608x968 cube (batch 1), "
                      "oracle/hyperpri_oracle.py on torch-CPU (Conv3d first layer as the reference), "
                      f"{t:.2f} s/step"}


# BASELINE.json's other configurations, per GPU: name -> (constructor, input shape for batch b, per-GPU batch,
# algorithmic GFLOP per unit fwd+bwd from SURVEY.md 8d, what BASELINE.json calls it)
OTHER_CONFIGS = {
    "C1_unet3_rgb": (lambda HP: HP.UNet(3, 1, bilinear=False), lambda b: (b, 3, 608, 968), 2, 2591.51,
                     "UNET n_channels=3 RGB 608x968 batch=2 (configs[0]; the reference runs it on CPU)"),
    "C3_spectralunet_1650": (lambda HP: HP.SpectralUNET(238, 1, 1650), lambda b: (b, 238, 608, 700), 1, 77150.90,
                             "SpectralUNET n_channels=238 spectral_bn_size=1650 patch 608x700, no model shard (configs[2]); per-GPU batch 1"),
    "C5_cubenet128_300": (lambda HP: HP.CubeNET(300, 1, first_depth=128, bilinear=False), lambda b: (b, 1, 300, 608, 968), 2, 3986.84,
                          "CubeNET-128 full-band 300-ch 968x608 (configs[4]; per-GPU share of the 8-GPU DDP job, batch 2)"),
}


def run_other_configs(dev, modes_for, steps=10, settle_s=1.0):
    """One GPU, forward + BCE loss + backward (the metric's step) per (config, mode), timed as the 238->64 leg is: back-to-back steps
    for ``settle_s`` seconds (at least 3: the allocator meets the shapes, the chip reaches the clock it holds under this load), then
    ``steps`` timed steps -- ms/step, units/s, model TFLOP/s (SURVEY.md 8d's algorithmic flops), peak memory, device allocations inside
    the timed steps -- and one more step under the per-kernel HIP-event log for the mode's dominant MFMA kernel (``roofline``)."""
    import gc
    import hyperpri_amd as HP
    from hyperpri_amd import engine
    out = {}
    for name, (mk, shp, batch, gflop, what) in OTHER_CONFIGS.items():
        row = {"workload": what, "batch": batch}
        for mode in modes_for(name):
            net = mk(HP).to(dev).train()
            synth_init_(net)
            HP.set_precision(net, mode)
            x = torch.empty(shp(batch), dtype=torch.float32, device=dev)
            mask = torch.empty((batch, 1) + tuple(x.shape[-2:]), dtype=torch.float32, device=dev)
            for i in range(batch):
                engine.synth_fill_(x[i], 1234 + i, mode=0)
                engine.synth_fill_(mask[i], 4321 + i, mode=1, thr=0.9)

            def step():
                for p in net.parameters():
                    p.grad = None
                _, loss = HP.forward_loss(net, x, mask)
                loss.backward()
                return loss
            torch.cuda.reset_peak_memory_stats(dev)
            t0, nwarm = time.perf_counter(), 0
            while nwarm < 3 or time.perf_counter() - t0 < settle_s:
                step()
                nwarm += 1
                if nwarm % 2 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            m0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            row[mode] = {"ms_per_step": round(dt * 1e3, 2), "units_per_s": round(batch / dt, 3),
                         "model_tflops": round(batch * gflop / dt / 1e3, 1), "loss": round(float(loss.detach()), 6),
                         "settle_steps": nwarm, "timed_steps": steps,
                         "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                         "device_mallocs_in_timed_steps": int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - m0)}
            # the mode's dominant MFMA kernel against its roofline (HIP events on the launching stream, one extra step)
            engine.enable_event_log(True)
            step()
            torch.cuda.synchronize()
            summ = engine.event_log_summary()
            engine.enable_event_log(False)
            if summ:
                tag, v = max(summ.items(), key=lambda kv: kv[1]["total_ms"])
                peak = PEAK_F32_MFMA_TFLOPS if mode == "fp32" else 2500.0
                row[mode]["roofline"] = {"bound": "mfma", "kernel": tag, "achieved": round(v["executed_tflops"], 1), "peak": peak, "unit": "TFLOP/s",
                                         "frac": round(v["executed_tflops"] / peak, 4), "avg_launch_ms": round(v["avg_ms"], 4),
                                         "launches_per_step": v["launches"], "ms_per_step": round(v["total_ms"], 3),
                                         "algorithmic_gflop_per_launch": round(v["flops_per_launch"] / 1e9, 2),
                                         "traffic": config_traffic(name, mode, tag)}
            del net, x, mask, step, loss
            gc.collect()
            torch.cuda.empty_cache()
        out[name] = row
    return out


def config_traffic(name, mode, tag):
    """HBM bytes per launch of a side config's dominant kernel, REPLAYED from profiles/rNN_<config>_<mode>_pmc_traffic.json (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of tools/run_config.py on the same build: tools/pmc_c3.sh + tools/pmc_traffic.py), or None."""
    import glob
    key = {"C3_spectralunet_1650": "c3"}.get(name)
    if key is None:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{key}_{mode}_pmc_traffic.json")))
    if not files:
        return None
    kern, why = _replayable(files[-1])
    if kern is None:
        return None
    want = "gemm_bf16v3_kernel<0>" if "gemm" in tag or "linear" in tag.lower() else None
    for k, v in kern.items():
        if want and want in k:
            return round(v["hbm_bytes_per_launch"])
    return None


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """``python bench.py --gpus N`` with no launcher around it: start N ranks with torch.distributed.run as a CHILD
    process (this parent never touches the GPU, so nothing is exec'ed over an initialised device), relay rank 0's JSON
    line and exit with the child's code.  Lightning does the same for ``strategy="ddp"`` (PLTrainer.py:434-442)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(f"bench.py: a rank failed (torch.distributed.run exit code {rc})" if rc > 0 else rc)
    if line is None:
        raise SystemExit("bench.py: the ranks finished without printing a result line")
    print(line, flush=True)


def trace(msg: str) -> None:
    """HPRI_BENCH_TRACE=1: leg-by-leg progress on stderr (one line per rank and leg) -- where a multi-rank run stopped, if it did."""
    if os.environ.get("HPRI_BENCH_TRACE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def dry_launch_rank():
    """Launcher rehearsal on CPU (tests/test_bench_launcher.py): gloo ranks, the contract's barrier + max-over-ranks
    timing around a trivial all-reduce, one JSON line from rank 0.  No GPU work, no claims."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("HPRI_DRY_LAUNCH_FAIL_RANK") == str(rank):     # tests: a rank that dies before the rendezvous
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    t = torch.ones(4) * (rank + 1)
    dist.all_reduce(t)
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launcher rehearsal (no GPU work)", "value": None, "n_gpus": world,
                          "allreduce_sum": float(t[0]), "seconds": float(dt), "ranks": world, "global_batch": world * BATCH,
                          "sample_seeds_last_rank": [1234 + (world - 1) * BATCH + i for i in range(BATCH)]}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-optimizer-leg", action="store_true",
                    help="skip the optimizer-step comparison (its torch.optim.Adam half is the only ATen work of a bench run; "
                         "profiling passes use this so that their kernel tables show the hot path alone)")
    ap.add_argument("--no-predict-leg", action="store_true", help="skip the forward-only (eval, inference_mode) leg")
    ap.add_argument("--no-training-shaped", action="store_true",
                    help="skip the `value_training_shaped` leg (the same loop with FusedAdam.step() after every backward)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` leg (BASELINE.json's other configurations C1 / C3 / C5 per GPU, fp32 and bf16, 3 + 3 steps each)")
    ap.add_argument("--bf16-steps", type=int, default=10,
                    help="extra steps in each of the precision modes bf16x6, bf16x3 and bf16 (reported as 'bf16x6_mode' / 'bf16x3_mode' / "
                         "'bf16_mode', never as 'value'); 0 = skip")
    ap.add_argument("--settle-seconds", type=float, default=0.0,
                    help="upper bound of an untimed settle phase in front of the warm-up steps (0 = none, the default since round 3: "
                         "five driver-shaped sequences -- GPU tests, then this script -- with engine.throttle alone gave 62.37-62.64 "
                         "cubes/s, none slow: profiles/r03_no_settle_sequences.txt); see the comment at the loop")
    ap.add_argument("--force-sync", action="store_true",
                    help="rehearsal: run the RCCL GradSync path (process group, hooks, all-reduce) even with one rank")
    ap.add_argument("--rccl-max-channels", type=int, default=0,
                    help="N > 1: cap RCCL's channels (NCCL_MAX_NCHANNELS, set before the communicator is created): every channel is a workgroup "
                         "that holds part of a compute unit while a bucket's all-reduce runs beside the backward (tools/cu_share_probe.py); 0 = RCCL's default")
    ap.add_argument("--stock-ddp", action="store_true",
                    help="with --force-sync at N = 1: also run the stock-DistributedDataParallel leg that every N > 1 run takes")
    ap.add_argument("--no-stock-ddp", action="store_true", help="N > 1: skip the stock-DistributedDataParallel leg")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher rehearsal: the ranks run a gloo all-reduce on the CPU instead of the GPU workload")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, sys.argv[1:])       # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_launch:
        return dry_launch_rank()
    # main, weight-gradient, RCCL and hand-over streams each want their own hardware queue (engine.SIDE_STREAM_WITH_SINK);
    # the runtime reads this when it initialises, i.e. at the first HIP call below
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # HPRI_BENCH_ONE_GPU=1: rehearsal of the N > 1 control flow on a ONE-GPU box -- every rank computes on cuda:0 and the
    # collective runs over gloo (RCCL refuses two ranks on one device).  Not a measurement: the line says so.
    one_gpu = os.environ.get("HPRI_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_pg = world > 1 or args.force_sync
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.rccl_max_channels > 0:
            os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_max_channels)
        # librccl prints a version banner to fd 1 when the communicator comes up: stdout must carry the ONE result line only
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if one_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()                        # communicator creation (eager with device_id, but make sure)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import hyperpri_amd as HP
    from hyperpri_amd import engine
    from hyperpri_amd.ddp import GradSync

    net = HP.CubeNET(BANDS, 1, first_depth=64, bilinear=False).to(dev).train()
    synth_init_(net)
    x = torch.empty((BATCH, 1, BANDS, H, W), dtype=torch.float32, device=dev)
    mask = torch.empty((BATCH, 1, H, W), dtype=torch.float32, device=dev)
    for i in range(BATCH):
        n = rank * BATCH + i                        # global sample index -> seeds 1234+n / 4321+n
        engine.synth_fill_(x[i], 1234 + n, mode=0)
        engine.synth_fill_(mask[i], 4321 + n, mode=1, thr=0.9)
    crit = HP.BCEWithLogitsLoss()        # nn.BCEWithLogitsLoss() semantics on the HIP path (csrc/step.hip)
    fused_loss = os.environ.get("HPRI_BENCH_FUSED_LOSS", "1") != "0"
    # N > 1: this rank's gradients from the PLAIN loop (no sink, no collective) before the gradient sink is installed: after the
    # timed steps every reduced gradient is compared with the mean of the ranks' plain gradients (`grad_sync.max_rel_err`; the
    # kernels are deterministic and the parameters frozen, so over gloo the two are bit-equal and over RCCL equal to its
    # summation order) -- ranks that merely hold EQUAL gradients say nothing about a stale slab or an early hand-over
    plain_flat = None
    if use_pg and world > 1:
        crit(net(x), mask).backward()
        torch.cuda.synchronize()
        plain_flat = torch.cat([p.grad.detach().reshape(-1) for p in net.parameters()])
        for p in net.parameters():
            p.grad = None
    sync = GradSync(net, force=args.force_sync) if use_pg else None

    def step():
        for p in net.parameters():
            p.grad = None
        if fused_loss:
            _, loss = HP.forward_loss(net, x, mask)     # same loss; computed inside the head's kernels
        else:
            loss = crit(net(x), mask)
        loss.backward()
        if sync is not None:
            sync.finish()
        return loss

    def fence():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- settle (OFF by default since round 3; kept as an option).  Round 2's finding: twice the first process on a box that had just run the GPU
    #      test suite measured 190-200 ms/step in the timed loop while every later leg of the same process, and single FENCED
    #      steps before it, ran at the normal 32 ms: the slowness belongs to the first seconds of back-to-back (host running
    #      ahead) steps of a fresh process, i.e. to the caching allocator growing its pool while blocks are still held by queued
    #      work.  So the settle phase runs what the measurement runs -- bursts of back-to-back steps, fenced only at their ends --
    #      until two bursts in a row are within 8 % of the fastest burst seen, for at most --settle-seconds; the number of bursts
    #      is agreed across ranks (every step holds collectives).  Allocator counters before / after are reported.
    settle = {"bursts": 0}
    if args.settle_seconds > 0 and not one_gpu:       # (the one-GPU gloo rehearsal takes 44 s per step: no settle there)
        burst = max(1, min(args.steps, 5))
        ms0 = torch.cuda.memory_stats(dev)
        seen = []
        t_begin = time.perf_counter()
        while True:
            fence()
            s0 = time.perf_counter()
            for _ in range(burst):
                step()
            fence()
            seen.append((time.perf_counter() - s0) / burst)
            calm = len(seen) >= 2 and max(seen[-2:]) <= 1.08 * min(seen)
            stop = calm or (time.perf_counter() - t_begin) > args.settle_seconds or len(seen) >= 200
            if use_pg:                          # (also with one rank under --force-sync, so that this path runs on a one-GPU box)
                flag = torch.tensor([0.0 if stop else 1.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                stop = float(flag.item()) == 0.0
            if stop:
                break
        ms1 = torch.cuda.memory_stats(dev)
        settle = {"bursts": len(seen), "steps_per_burst": burst, "ms_per_step": [round(t * 1e3, 2) for t in seen[:6]] + (["..."] if len(seen) > 6 else []),
                  "device_mallocs": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                  "reserved_gib": round(ms1.get("reserved_bytes.all.current", 0) / 2 ** 30, 1)}
    trace("warm-up steps")
    for _ in range(args.warmup):
        step()
    fence()
    trace("timed steps")
    mallocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    settle["device_mallocs_in_timed_region"] = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - mallocs0)   # 0 = the pool was grown
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.detach())
    grad_sync = None
    if sync is not None:
        ov = sync.overlap_ms()           # last timed step: per bucket, all-reduce issue -> finish() return (stream time)
        equal = None
        max_rel_err = None
        if plain_flat is not None:
            ref = plain_flat.clone()
            dist.all_reduce(ref, op=dist.ReduceOp.SUM)
            ref.div_(world)
            got = torch.cat([p.grad.detach().reshape(-1) for p in net.parameters()])
            errs, off = [], 0
            for p in net.parameters():
                n = p.numel()
                d = (got[off:off + n].double() - ref[off:off + n].double()).abs().max()
                errs.append(d / ref[off:off + n].double().abs().max().clamp_min(1e-30))
                off += n
            mre = torch.stack(errs).max()
            dist.all_reduce(mre, op=dist.ReduceOp.MAX)
            max_rel_err = float(mre)
            del ref, got
        if one_gpu and world > 1:
            # rehearsal check: after finish() every rank must hold the same (averaged) gradients
            cs = torch.stack([torch.stack([p.grad.double().abs().sum(), p.grad.double().sum()]) for p in net.parameters()]).sum(0)
            lo, hi = cs.clone(), cs.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            equal = bool(torch.equal(lo, hi))
        grad_sync = {"backend": "gloo on one GPU (REHEARSAL, not a measurement)" if one_gpu else "nccl (RCCL)",
                     "max_rel_err": max_rel_err,
                     "max_rel_err_is": "max over parameter tensors and ranks of max|reduced gradient - mean of the ranks' plain-loop HIP "
                                       "gradients| / max|mean| (0.0 = bit-equal; null at N = 1)",
                     "ranks_hold_equal_gradients": equal, "grad_mb": round(sum(b.flat.numel() for b in sync.buckets) * 4 / 2 ** 20, 1),
                     "gradients_written_in_place": True, "overlap": ov, "buckets_issued_by": dict(sync.issued),
                     # what bounds the compute units the collective's kernels hold beside the backward (RCCL reads these when the
                     # communicator is created; unset = its own choice for the topology)
                     "rccl_channel_env": {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_NTHREADS", "RCCL_MSCCL_ENABLE")},
                     "item_queues": bool(engine.ITEM_QUEUE),
                     "note": "buckets are issued from inside the backward tape as their last gradient lands; "
                             "issue_to_finish_ms[0] is the window in which communication ran beside the rest of backward"}

    # ---- the same loop as a training run shapes it: FusedAdam.step() after every backward, so every parameter changes every
    #      step and the packed-weight caches (Winograd U, data-gradient panels) are REBUILT every step -- `value`'s loop keeps the
    #      parameters frozen and therefore pays 0 pack launches.  Reported beside `value`, never as it (SURVEY.md 8d excludes the
    #      optimizer from the metric).  All ranks take part (a step holds collectives); parameters are restored afterwards.
    training_shaped = None
    trace("timed steps done; training-shaped leg")
    if not args.no_training_shaped:
        with torch.no_grad():
            saved = [p.detach().clone() for p in net.parameters()]
        opt = HP.FusedAdam(net.parameters(), lr=1e-3)
        packs0 = engine.PACK_LAUNCHES
        for _ in range(4):                 # (the optimizer state and the re-packed panels grow the allocator's pool: a device
            step(); opt.step()             # allocation inside the timed loop costs ~40 ms on this stack, see `device_mallocs`)
        fence()
        packs1 = engine.PACK_LAUNCHES
        mall_t = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        tt = time.perf_counter()
        for _ in range(args.steps):
            loss_t = step(); opt.step()
        fence()
        dtt = time.perf_counter() - tt
        if world > 1:
            t = torch.tensor([dtt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtt = float(t.item())
        training_shaped = {"value": round(world * BATCH * args.steps / dtt, 4), "unit": "cubes/s", "steps": args.steps,
                           "ms_per_step": round(dtt / args.steps * 1e3, 3), "final_loss": round(float(loss_t.detach()), 6),
                           "pack_launches_per_step": (engine.PACK_LAUNCHES - packs1) / args.steps,
                           "device_mallocs": int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - mall_t),
                           "what": "forward + loss + backward" + (" + gradient all-reduce" if world > 1 else "") +
                                   " + FusedAdam(lr=1e-3).step(): weights change every step, packed copies rebuilt every step"}
        with torch.no_grad():
            for p, q in zip(net.parameters(), saved):
                p.copy_(q)
        del saved, opt
        engine.bump_param_epoch()
        step(); fence()                    # re-pack outside anything timed below

    # ---- optimizer step: excluded from the metric, reported beside it ----
    optimizer_step = None
    if rank == 0 and not args.no_optimizer_leg:
        def time_opt(opt, n=10):
            for _ in range(2):
                opt.step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                opt.step()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        with torch.no_grad():
            saved = [p.detach().clone() for p in net.parameters()]
        nparam = sum(p.numel() for p in net.parameters())
        ms_fused = time_opt(HP.FusedAdam(net.parameters(), lr=1e-3))
        ms_torch = time_opt(torch.optim.Adam(net.parameters(), lr=1e-3))
        with torch.no_grad():
            for p, q in zip(net.parameters(), saved):
                p.copy_(q)                         # the timed steps must not change the workload that follows
        del saved
        optimizer_step = {"optimizer": "Adam(lr=1e-3) over %d parameters in %d tensors" % (nparam, len(list(net.parameters()))),
                          "hip_multi_tensor_ms": round(ms_fused, 4), "torch_optim_adam_ms": round(ms_torch, 4),
                          "hbm_gb_s": round(7 * 4 * nparam / ms_fused / 1e6, 1),
                          "note": "4 reads + 3 writes of fp32 per element; not part of `value`"}

    # ---- secondary lines: the same workload in the other precision modes (never the headline) ----
    ms_step_hint = {"bf16": 0.010, "f16": 0.010, "bf16x3": 0.024, "bf16x6": 0.039}      # seconds per step, for sizing the warm-up (~0.25 s) only
    def timed_mode(mode):
        time.sleep(2.0)            # let the clocks recover from the previous mode (DVFS give-back), outside any timed region
        HP.set_precision(net, mode)
        # untimed: the caching allocator meets this mode's buffer shapes (a device allocation inside the timed steps costs ~40 ms),
        # and the chip, idle during the sleep above, is back at the clock it holds under this load (it rises over tens of
        # milliseconds: profiles/r04_first_conv_ramp.txt; 4 bf16 steps = 40 ms were not enough: 204.4 vs 207.9 cubes/s over 20 steps)
        for _ in range(1 if one_gpu else max(4, int(0.25 / max(ms_step_hint.get(mode, 0.02), 1e-3)))):     # (one-GPU rehearsal: ranks time-slice one chip, ~40 s per step)
            step()
        fence()
        m0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        tb = time.perf_counter()
        for _ in range(args.bf16_steps):
            lossb = step()
        fence()
        dtb = time.perf_counter() - tb
        if world > 1:
            t = torch.tensor([dtb], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtb = float(t.item())
        HP.set_precision(net, "fp32")
        return {"value": round(world * BATCH * args.bf16_steps / dtb, 4), "unit": "cubes/s", "steps": args.bf16_steps,
                "ms_per_step": round(dtb / args.bf16_steps * 1e3, 3), "loss": round(float(lossb.detach()), 6),
                "device_mallocs_in_timed_steps": torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - m0}

    bf16_mode = bf16x3_mode = bf16x6_mode = f16_mode = None
    trace("precision-mode legs")
    if args.bf16_steps > 0:
        bf16x6_mode = timed_mode("bf16x6")
        bf16x6_mode.update({
            "dtype": "f32 operands split exactly into 3 bf16 planes (24 mantissa bits), 6 x v_mfma_f32_32x32x16_bf16 per product "
                     "(the 3 cross terms below 2^-24 dropped), f32 accumulate",
            "parity": "fp32-class: error of a convolution against fp64 1.0-1.6e-6 vs 1.3-1.5e-6 for the fp32 MFMA path "
                      "(profiles/r01_precision_error.json); max |dlogit| 1.6e-5 vs the CPU reference (exact path: 1.1e-5), all 42 "
                      "held-out Dice/IoU equal to 4 dp (profiles/r01_bf16x6_dice_parity.json); reported separately because it "
                      "is an emulation of the fp32 product, not the fp32 MFMA instruction"})
        bf16x3_mode = timed_mode("bf16x3")
        bf16x3_mode.update({
            "dtype": "bf16 hi+lo operands (16 mantissa bits), 3 x v_mfma_f32_32x32x16_bf16 per product, f32 accumulate",
            "parity": "meets the fp32 contract on every fixture: full-size logits within 1e-3 of the reference, Dice/IoU equal "
                      "to 4 dp on all 42 held-out comparisons, max |dlogit| 7.8e-5 (profiles/r01_bf16x3_dice_parity.json); "
                      "reported separately because BASELINE config C2 names fp32"})
        f16_mode = timed_mode("f16")
        f16_mode.update({
            "dtype": "IEEE half operands / f32 accumulate (v_mfma_f32_16x16x32_f16 forward + data gradient, 32x32x16 weight gradient): the bf16 "
                     "mode's kernels with half as the 16-bit type (libhyperpri_hip_f16.so); activations, pre-BN tensors and activation gradients "
                     "stored as half, activation gradients under a power-of-two loss scale (2^21 for 2 x 608 x 968 logits)",
            "parity": "at identical weights on the benched shape: max |dlogit| 4.7e-3 against the reference fixture (bf16: 5.0e-2), 0.08 % sign "
                      "flips, loss within 3e-7, Dice / IoU within 1e-4; gradients against the reference's fp64 samples: worst tensor 0.21 "
                      "relative L2 / cosine 0.976 (bf16: 0.58 / 0.82) -- tests/test_gpu_f16.py, profiles/r05_f16_grad_parity_c2.json; "
                      "NOT the headline value"})
        bf16_mode = timed_mode("bf16")
        bf16_mode.update({
            "dtype": "bf16 operands / f32 accumulate (v_mfma_f32_16x16x32_bf16 forward + data gradient, 32x32x16 weight gradient); BN statistics, pooling, gradients f32",
            "parity": "Dice/IoU level only (held-out split emulation, 14 cubes x 3 variants vs the fp32 oracle: at identical weights max "
                      "|dlogit| 5.0e-2, <= 0.47 % sign flips; after 20 Adam steps run in this mode max |dlogit| 0.19 with the same training "
                      "losses to 2e-3; |dDice| <= 1.6e-4 everywhere: profiles/r04_bf16_dice_parity.json, final round-4 build -- activations, "
                      "pre-BN tensors and single-reader / skip gradients stored as bf16); NOT the headline value"})

    # ---- the predict path (PLTrainer.py:530-532: eval mode under inference_mode) on the same cubes: forward only, eval-mode BatchNorm
    #      folded into the convolutions; N = 1 only (it holds no collective); never part of `value` ----
    predict = None
    if rank == 0 and world == 1 and not args.no_predict_leg:
        trace("predict leg")
        net.eval()
        predict = {"what": "CubeNET-64 forward only, eval mode under torch.inference_mode(), the same two cubes; cubes/s per mode"}
        for mode in ("fp32", "bf16", "f16"):
            HP.set_precision(net, mode)
            with torch.inference_mode():
                for _ in range(5):
                    net(x)
                torch.cuda.synchronize()
                npred = 20 if mode == "fp32" else 50
                tp0 = time.perf_counter()
                for _ in range(npred):
                    net(x)
                torch.cuda.synchronize()
                dtp = time.perf_counter() - tp0
            predict[mode] = {"value": round(BATCH * npred / dtp, 2), "unit": "cubes/s", "ms_per_forward": round(dtp / npred * 1e3, 3)}
        HP.set_precision(net, "fp32")
        net.train()
        torch.cuda.synchronize()

    roofline = None
    trace("per-kernel event pass")
    if not args.no_roofline:
        # per-kernel HIP events on the launching stream over 2 extra steps (events perturb the timing
        # slightly, so they are kept out of the headline region).  EVERY rank takes the two steps -- a step holds the gradient
        # collectives, and a rank stepping alone would wait for its peers for ever (tests/test_bench_rank_symmetry.py) -- rank 0 logs.
        engine.enable_event_log(rank == 0)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        summ = engine.event_log_summary()
        engine.enable_event_log(False)
    if rank == 0 and not args.no_roofline:
        dom = max(summ.items(), key=lambda kv: kv[1]["total_ms"])
        traffic, traffic_src = pmc_traffic(dom[0])
        wino = "winograd" in dom[0]
        roofline = {"bound": "mfma", "kernel": dom[0],
                    # flops the kernel ISSUES on the fp32 matrix pipe per launch / its HIP-event time: for the Winograd kernels
                    # 2 * tiles * 16 * Cin * Cout (16 multiplies per 2x2 outputs and channel pair), not the 36 of the direct sum
                    "achieved": round(dom[1]["executed_tflops"], 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(dom[1]["executed_tflops"] / PEAK_F32_MFMA_TFLOPS, 4),
                    "direct_equivalent_tflops": round(dom[1]["tflops"], 2),
                    "arithmetic": ("Winograd F(2x2,3x3) in fp32: `achieved`/`frac` count the multiply-adds executed (16 per 2x2 outputs "
                                   "and channel pair); `direct_equivalent_tflops` prices the same launches at SURVEY.md 8d's direct-"
                                   "convolution flops (36 per 2x2), which is what `value` x 2910.17 GFLOP/cube uses") if wino else
                                  "direct implicit GEMM: executed = algorithmic flops",
                    "traffic": traffic, "traffic_source": (traffic_src if (traffic_src or "").startswith("NOT REPLAYED") else
                                                           f"replayed from {traffic_src} (rocprofv3 --pmc passes of this workload on the "
                                                           "same build of the library, committed; not measured in this run)") if traffic_src else None,
                    "avg_launch_ms": round(dom[1]["avg_ms"], 4), "launches_per_step": dom[1]["launches"] // 2,
                    "algorithmic_gflop_per_launch": round(dom[1]["flops_per_launch"] / 1e9, 3),
                    "executed_gflop_per_launch": round(dom[1]["executed_flops_per_launch"] / 1e9, 3),
                    "all_mfma_kernels": {k: {"avg_ms": round(v["avg_ms"], 4), "tflops": round(v["tflops"], 2),
                                             "executed_tflops": round(v["executed_tflops"], 2),
                                             "launches_per_step": v["launches"] // 2,
                                             "ms_per_step": round(v["total_ms"] / 2, 3)} for k, v in sorted(summ.items())}}

    # ---- north_star's named kernel: the 238->64 encoder conv (models.py:169) against the bf16 MFMA roofline ----
    first_conv = None
    if rank == 0 and not args.no_roofline:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import first_conv as FC
        torch.cuda.empty_cache()
        r = FC.measure(reps=10, settle_s=1.0)
        fc_bytes, fc_src = first_conv_traffic()
        bp, b16 = r["bf16_planes"], r.get("bf16_planes_bf16_out")
        top = b16 or bp          # the form the bf16 step launches (engine.YR_BF16): bf16 planes in, bf16 pre-BN tensor out
        first_conv = {"layer": "CubeNET-64 first_conv 238->64, 3x3, batch 2, forward (bias + BN partial statistics in the epilogue)",
                      "mode": "bf16 operand planes resident in HBM, both operands by LDS-DMA, " + r.get("bf16_kernel", "") + ", f32 accumulate, "
                              + ("bf16 pre-BN output written from the accumulators -- the launch the bf16 step makes for this layer "
                                 "(SURVEY.md 7.3-2: bf16-in / bf16-out); `f32_out` = the same kernel writing an f32 output" if b16 else "f32 output")
                              + " (the layout pass that writes the input planes is a separate kernel); operands: the workload's synthetic "
                                "cube u in [0,1) and default-bound weights",
                      "timing": "HIP events over 50 back-to-back launches after 1 s of back-to-back launches of the same kernel (the state the "
                                "kernel is in inside a step); `*_burst_from_idle`: 10 launches right after three warm-up launches on an idle "
                                "chip, the protocol of rounds 1-3 (the clock is still rising: tools/first_conv.py, profiles/r04_first_conv_ramp.txt)",
                      "ms": top["ms"], "TF": top["tflops"], "frac_of_2.5PF": top["frac_of_2.5PF"],
                      "ms_burst_from_idle": top["ms_burst_from_idle"],
                      "hbm_bytes_algorithmic": int(top["algorithmic_hbm_mb"] * 1e6),
                      "hbm_bytes_measured": fc_bytes, "hbm_bytes_measured_source": fc_src,
                      "f32_out": {"ms": bp["ms"], "TF": bp["tflops"], "frac_of_2.5PF": bp["frac_of_2.5PF"],
                                  "hbm_bytes_algorithmic": int(bp["algorithmic_hbm_mb"] * 1e6),
                                  "ms_burst_from_idle": bp["ms_burst_from_idle"], "TF_burst_from_idle": bp["tflops_burst_from_idle"]},
                      "energy_floor": "profiles/r04_energy_floor.jsonl: a bare loop with this kernel's MFMA / ds_read / LDS-DMA mix per stage and "
                                      "no epilogue sustains 1297-1300 TF = 0.52 of 2.5 PF at 1.54-1.55 GHz in-kernel on random operands",
                      "fp32_kernel_same_layer": r["fp32"], "fp32_winograd_same_layer": r.get("fp32_winograd")}
        # the same launch in the f16 mode's library (IEEE half planes / weights / pre-BN output, v_mfma_f32_16x16x32_f16): the dtype
        # north_star names for the roofline claim
        rh = FC.measure(reps=10, settle_s=1.0, modes=("bf16_planes",), kind="f16")
        th = rh.get("bf16_planes_bf16_out") or rh["bf16_planes"]
        first_conv["f16"] = {"mode": "fp16 operand planes in, fp16 pre-BN output, hpri_conv_bf16v3 of libhyperpri_hip_f16.so (the launch of the f16 step)",
                             "ms": th["ms"], "TF": th["tflops"], "frac_of_2.5PF": th["frac_of_2.5PF"], "ms_burst_from_idle": th["ms_burst_from_idle"],
                             "f32_out": {"ms": rh["bf16_planes"]["ms"], "TF": rh["bf16_planes"]["tflops"],
                                         "frac_of_2.5PF": rh["bf16_planes"]["frac_of_2.5PF"]}}

    # ---- BASELINE.json's other configurations on this GPU (N = 1 only: they are per-GPU figures; never part of `value`) ----
    configs = None
    if rank == 0 and world == 1 and not args.no_configs:
        del x, mask
        net = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        configs = run_other_configs(dev, lambda name: ("fp32", "bf16x3", "bf16", "f16") if name.startswith("C3") else ("fp32", "bf16", "f16"))
        configs["note"] = ("per-GPU step = forward + BCEWithLogits + backward on synthetic inputs; per leg >= 1 s of back-to-back steps, then 10 timed "
                           "steps, then one step under the per-kernel event log (roofline of the leg's dominant MFMA kernel); fp32 = exact "
                           "fp32 MFMA (Winograd for 3x3), bf16x3 = 2 bf16 planes per operand (meets the fp32 contract: logits within 1e-3, "
                           "Dice/IoU to 4 dp on the reference fixture), bf16 = Dice-level parity; model_tflops = SURVEY.md 8d flops / time")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    # ---- the reference's own N > 1 caller: the same step under STOCK torch DistributedDataParallel (Lightning strategy="ddp",
    #      PLTrainer.py:434-442) instead of the GradSync sink -- the network runs as a chain of autograd nodes over one tape
    #      (hyperpri_amd/autograd.py::run_staged) and the reducer's buckets leave during backward.  Reported beside `value`, never
    #      as it.  Every rank takes part; the sink is removed first (one gradient consumer at a time).
    def stock_ddp_leg():
        nonlocal sync
        from torch.nn.parallel import DistributedDataParallel as DDP
        from hyperpri_amd import autograd as HA
        sync.remove()
        sync = None
        for p in net.parameters():
            p.grad = None
        ddp_net = DDP(net, device_ids=[dev_index], broadcast_buffers=False, gradient_as_bucket_view=True)

        def ddp_step():
            for p in net.parameters():
                p.grad = None
            lo = crit(ddp_net(x), mask)
            lo.backward()
            return lo
        stock_ddp = {"wrapper": "torch.nn.parallel.DistributedDataParallel(net, device_ids=[rank's GPU], broadcast_buffers=False, "
                                "gradient_as_bucket_view=True), 25 MiB buckets, backend " + ("gloo (REHEARSAL)" if one_gpu else "nccl (RCCL)"),
                     "what": "forward + BCEWithLogits + backward under stock DDP; value = cubes of all ranks / max-over-ranks time"}
        for mode in ("bf16", "fp32"):
            HP.set_precision(net, mode)
            for _ in range(1 if one_gpu else 6):          # (the reducer rebuilds its buckets in arrival order after the first step)
                ddp_step()
            fence()
            nst = 1 if one_gpu else args.steps
            tb = time.perf_counter()
            for _ in range(nst):
                lo = ddp_step()
            fence()
            dts = time.perf_counter() - tb
            if world > 1:
                t = torch.tensor([dts], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dts = float(t.item())
            stock_ddp[mode] = {"value": round(world * BATCH * nst / dts, 4), "unit": "cubes/s", "steps": nst,
                               "ms_per_step": round(dts / nst * 1e3, 3), "loss": round(float(lo.detach()), 6)}
        stock_ddp["autograd_nodes_stages"] = list(HA.LAST_PLAN)
        if plain_flat is not None:                       # fp32 ran last: its reduced gradients against the mean of the ranks' plain ones
            ref = plain_flat.clone()
            dist.all_reduce(ref, op=dist.ReduceOp.SUM)
            ref.div_(world)
            got = torch.cat([p.grad.detach().reshape(-1) for p in net.parameters()])
            errs, off = [], 0
            for p in net.parameters():
                n = p.numel()
                d = (got[off:off + n].double() - ref[off:off + n].double()).abs().max()
                errs.append(d / ref[off:off + n].double().abs().max().clamp_min(1e-30))
                off += n
            mre = torch.stack(errs).max()
            dist.all_reduce(mre, op=dist.ReduceOp.MAX)
            stock_ddp["max_rel_err_vs_mean_of_plain_gradients"] = float(mre)
            del ref, got
        del ddp_net
        for p in net.parameters():
            p.grad = None
        return stock_ddp

    trace("rank-0 legs done; closing barrier")
    if use_pg:
        dist.barrier()
    out = None
    if rank == 0:
        cubes = world * BATCH * args.steps
        value = cubes / dt
        out = {
            "metric": "HSI cubes/sec (608x968x238) fwd+bwd", "value": round(value, 4), "unit": "cubes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # the arithmetic type of `value`: fp32 MFMA unless the caller switched the package default (HPRI_PRECISION), which
            # the line then says instead of claiming f32
            "dtype": {"fp32": "f32", "bf16": "bf16 (f32 accumulate)", "bf16x3": "bf16x3 (2 bf16 planes per operand, f32 accumulate)",
                      "bf16x6": "bf16x6 (3 bf16 planes per operand: f32 emulation)"}[engine.DEFAULT_PRECISION],
            "data": "synthetic",
            "config": {"workload": "CubeNET-64 n_channels=238 HSI 608x968 " + engine.DEFAULT_PRECISION + ", per-GPU batch 2, train mode, "
                                   "BCEWithLogits, fwd+bwd" + (" + RCCL grad all-reduce" if world > 1 else ""),
                       "global_batch": world * BATCH, "parallelism": f"dp{world}"},
            "loss": round(loss_val, 6), "loss_in_head_kernels": fused_loss,
            "settle": settle,
            "rccl_ranks": world if use_pg else 0,
            "grad_sync": grad_sync,
            "model_tflops": round(value * GFLOP_PER_CUBE / 1e3 / world, 2),
            "model_tflops_note": "value x 2910.17 GFLOP/cube (direct-convolution flops of the reference graph, SURVEY.md 8d) per GPU; "
                                 "the fp32 path executes fewer multiplies than that (Winograd), so this can exceed the fp32 MFMA peak",
            "value_training_shaped": training_shaped,
            "roofline": roofline, "roofline_238to64": first_conv, "cpu_baseline": cpu, "optimizer_step": optimizer_step, "configs": configs, "bf16x6_mode": bf16x6_mode, "bf16x3_mode": bf16x3_mode, "bf16_mode": bf16_mode, "f16_mode": f16_mode,
            "predict": predict,
            "stock_ddp": None,
        }
    # The stock-DDP leg runs LAST, when everything the contract asks for is measured and the line is assembled: it is the one part of an
    # N > 1 run that no one-GPU box could rehearse over RCCL with more than one rank.  If it raises, or has not come back within two
    # minutes (a rank that failed alone leaves its peers in a collective), rank 0 prints the line without it and every rank leaves.
    if use_pg and (world > 1 or args.stock_ddp) and not args.no_stock_ddp and net is not None:
        trace("stock DistributedDataParallel leg")

        def give_up():
            if rank == 0:
                out["stock_ddp"] = {"error": "the leg did not finish within 120 s; the line above it is complete"}
                print(json.dumps(out), flush=True)
            os._exit(0)
        dog = threading.Timer(3600.0 if one_gpu else 120.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            res = stock_ddp_leg()
        except Exception as e:                   # noqa: BLE001 -- whatever it is, the measured line must still go out
            res = {"error": repr(e)[:500]}
        dog.cancel()
        if rank == 0:
            out["stock_ddp"] = res
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
