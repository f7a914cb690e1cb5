"""One rank of tests/test_gpu_ddp_stock.py (started as a fresh child process; not collected by pytest).

    python tests/_ddp_stock_rank.py <rank> <world> <port> <out.npz> <backend> <size> [precision]

The reference's own multi-GPU caller: Lightning strategy="ddp" (PLTrainer.py:434-442) wraps the network in STOCK
torch.nn.parallel.DistributedDataParallel.  Here the network is wrapped the same way (no GradSync, no sink); under the process
group its tape runs as a chain of autograd nodes (hyperpri_amd/autograd.py: run_staged), so the reducer gets the gradients of the
decoder and the bottleneck while the encoder's backward is still being enqueued.  A communication hook records, per bucket, its
position in the sequence of C-ABI launches and a HIP event on the compute stream.

``size``: "tiny" = CubeNET(6,1,64) on 2 x 36x50 cubes (two ranks on cuda:0 over gloo: RCCL refuses two ranks on one device);
"full" = the benched shape, CubeNET(238,1,64) on 2 x 608x968 cubes (one rank over RCCL)."""
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, out, backend, size = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    prec = sys.argv[7] if len(sys.argv) > 7 else "fp32"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import hyperpri_amd as H
    from hyperpri_amd import _lib, autograd
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from torch.nn.parallel import DistributedDataParallel as DDP
    import bench

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if size == "full":
        net = H.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
        bench.synth_init_(net)
        x = torch.empty(2, 1, 238, 608, 968, device=dev)
        m = torch.empty(2, 1, 608, 968, device=dev)
        for n in range(2):
            H.engine.synth_fill_(x[n], 1234 + 2 * rank + n)
            H.engine.synth_fill_(m[n], 4321 + 2 * rank + n, mode=1, thr=0.9)
    else:
        from oracle import hyperpri_oracle as O        # inputs only: the counter-based generator
        net = H.CubeNET(6, 1, first_depth=64, bilinear=False)
        shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
        net.load_state_dict(O.synth_state_dict(shapes))
        net = net.to(dev).train()
        u = lambda seed, shape: torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())
        x = u(1235 + rank, (2, 1, 6, 36, 50)).to(dev)
        m = (u(4321 + rank, (2, 1, 36, 50)) > 0.9).float().to(dev)
    if prec != "fp32":
        H.set_precision(net, prec)
    lossf = torch.nn.BCEWithLogitsLoss()
    res = {}
    # ---- this rank's plain step: no process group yet, so the network is ONE autograd node ----
    logits = net(x)
    lossf(logits, m).backward()
    torch.cuda.synchronize()
    assert autograd.LAST_PLAN == []
    res["plain_logits_head"] = logits.detach().reshape(-1)[:4096].cpu().numpy()
    plain = {}
    for k, p in net.named_parameters():
        plain[k] = p.grad.detach().clone()
        p.grad = None
    sd = {k: v.clone() for k, v in net.state_dict().items()}      # (BN buffers moved: every later step starts from here)

    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ddp = DDP(net, device_ids=[0], broadcast_buffers=False)

    log = []                          # ("launch", name) | ("bucket", index, bytes, is_last)
    real_call = _lib.call

    def spy(name, *args):
        log.append(("launch", name))
        return real_call(name, *args)
    _lib.call = spy
    events = []

    def hook(state, bucket):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        events.append(ev)
        log.append(("bucket", bucket.index(), bucket.buffer().numel() * 4, bool(bucket.is_last())))
        return default_hooks.allreduce_hook(state, bucket)
    ddp.register_comm_hook(None, hook)

    steps = 3                         # (the reducer re-buckets the parameters in arrival order after the first step)
    for step in range(steps):
        net.load_state_dict(sd)
        for p in net.parameters():
            p.grad = None
        del log[:], events[:]
        logits = ddp(x)
        loss = lossf(logits, m)
        t0 = torch.cuda.Event(enable_timing=True)
        t0.record()
        mark = len(log)
        loss.backward()
        t1 = torch.cuda.Event(enable_timing=True)
        t1.record()
        torch.cuda.synchronize()
    _lib.call = real_call
    res["plan"] = np.asarray(autograd.LAST_PLAN)
    res["segmented_logits_head"] = logits.detach().reshape(-1)[:4096].cpu().numpy()
    bwd = log[mark:]
    launches = [i for i, e in enumerate(bwd) if e[0] == "launch"]
    wg = [i for i, e in enumerate(bwd) if e[0] == "launch" and "wgrad" in e[1] and "plan" not in e[1]]
    buckets = [(i, e) for i, e in enumerate(bwd) if e[0] == "bucket"]
    res["n_launches"] = len(launches)
    res["last_wgrad_pos"] = wg[-1]
    res["bucket_pos"] = np.asarray([i for i, _ in buckets])
    res["bucket_bytes"] = np.asarray([e[2] for _, e in buckets], dtype=np.int64)
    res["bucket_is_last"] = np.asarray([int(e[3]) for _, e in buckets])
    # launches of the backward still to be enqueued when each bucket was handed to the communication hook
    res["launches_after_bucket"] = np.asarray([sum(1 for j in launches if j > i) for i, _ in buckets])
    total = t0.elapsed_time(t1)
    res["backward_ms"] = total
    res["bucket_ms"] = np.asarray([t0.elapsed_time(ev) for ev in events], dtype=np.float64)
    for k, p in net.named_parameters():
        res["g/" + k] = p.grad.detach().cpu().numpy()
        res["plain/" + k] = plain[k].cpu().numpy()
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
