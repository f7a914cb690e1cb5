#!/usr/bin/env python3
"""What a communication kernel that holds compute units costs the step.  The persistent MFMA kernels start
2 x CUs workgroups, two per CU, with 72-78 KB of LDS and all 512 vector registers of a SIMD pair: on a CU where an RCCL workgroup
sits only one of them fits, the other waits for a retirement.  With FIXED item lists that late workgroup then walks its whole list
alone (a second wave: up to 2x the launch); with the per-XCD item counters every resident workgroup draws the next item and a late
one finds the queue empty.  Here a hog (tools/cu_hog.hip: W workgroups x 512 threads x 64 KB LDS, asleep until a deadline) holds W
CUs on a second stream for the length of several steps; the step time with and without it is recorded for the fp32 step, the bf16
step and a C3-shaped bf16 step.
usage: cu_share_probe.py [--fixed-lists] [--reserve=N] [out.json] [label]"""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine  # noqa: E402

HOG = os.path.join(ROOT, "tools", "bin", "libcuhog.so")
if not os.path.exists(HOG) or os.path.getmtime(HOG) < os.path.getmtime(os.path.join(ROOT, "tools", "cu_hog.hip")):
    os.makedirs(os.path.dirname(HOG), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", HOG,
                           os.path.join(ROOT, "tools", "cu_hog.hip")])
hog = ctypes.CDLL(HOG)
hog.cu_hog_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
hog.cu_hog_launch_lds.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
hog_stream = torch.cuda.Stream(device=dev)
sink = torch.zeros(1024, dtype=torch.int32, device=dev)


def make(kind, prec):
    if kind == "c2":
        net = HP.CubeNET(238, 1, first_depth=64, bilinear=False)
        shape, hw, n = (2, 1, 238, 608, 968), (608, 968), 2
    else:                       # C3's layer widths on a quarter of its pixels (the full shape turns over 80 GiB per step)
        net = HP.SpectralUNET(238, 1, 1650)
        shape, hw, n = (1, 238, 304, 350), (304, 350), 1
    net = net.to(dev).train()
    bench.synth_init_(net)
    HP.set_precision(net, prec)
    x = torch.empty(shape, device=dev)
    m = torch.empty((n, 1) + hw, device=dev)
    for i in range(n):
        engine.synth_fill_(x[i], 1234 + i)
        engine.synth_fill_(m[i], 4321 + i, mode=1, thr=0.9)
    crit = torch.nn.BCEWithLogitsLoss()

    def step():
        for p in net.parameters():
            p.grad = None
        crit(net(x), m).backward()
    return step


def timed(step, n, hog_wg, hog_ms, lds=65536):
    """ms per step over n back-to-back steps; with hog_wg > 0 a hog of that many workgroups is resident for hog_ms from just
    before the first step (the steps must end inside that window: checked by the caller against the returned time)."""
    torch.cuda.synchronize()
    if hog_wg:
        with torch.cuda.stream(hog_stream):
            assert hog.cu_hog_launch_lds(hog_wg, float(hog_ms), lds, sink.data_ptr(), hog_stream.cuda_stream) == 0
        time.sleep(0.002)                   # the hog is on the chip before the first kernel of the step is enqueued
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    args = [a for a in sys.argv[1:] if a != "--fixed-lists" and not a.startswith("--reserve=")]
    if "--fixed-lists" in sys.argv[1:]:
        engine.ITEM_QUEUE = False           # the round-4 behaviour: no item queues are registered, workgroups walk fixed lists
    reserve = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--reserve=")]
    if reserve:
        engine.set_plan_option("wgrad_cu_reserve", reserve[0])      # the fp32 Winograd weight gradient plans for that many fewer CUs
    out_path = args[0] if len(args) > 0 else None
    label = args[1] if len(args) > 1 else "current build"
    res = {"label": label, "library_stamp": bench._lib_stamp(), "hog": "W workgroups x 512 threads x 64 KB LDS, asleep (tools/cu_hog.hip)",
           "cus": torch.cuda.get_device_properties(0).multi_processor_count, "wgrad_cu_reserve": (reserve[0] if reserve else 0), "legs": {}}
    for name, kind, prec, nsteps in (("fp32_c2", "c2", "fp32", 4), ("bf16_c2", "c2", "bf16", 10), ("bf16_c3_quarter", "c3", "bf16", 4)):
        step = make(kind, prec)
        for _ in range(4):
            step()
        base = sorted(timed(step, nsteps, 0, 0) for _ in range(3))[1]
        leg = {"ms_per_step_alone": round(base, 3), "with_hog": {}}
        for w in (8, 16, 32, 64):
            window = base * nsteps * 2.6 + 20.0          # covers a 2x slowdown with room to spare; the hog leaves at its deadline anyway
            ts = sorted(timed(step, nsteps, w, window) for _ in range(3))
            t = ts[1]
            leg["with_hog"][str(w)] = {"ms_per_step": round(t, 3), "slowdown": round(t / base, 4), "cu_share": round(w / res["cus"], 4),
                                       "excess_over_cu_share": round(t / base - 1.0 - w / res["cus"], 4),
                                       "steps_ended_inside_hog_window": bool(t * nsteps < window)}
            print(name, "W", w, leg["with_hog"][str(w)], flush=True)
        # controls: ONE hog workgroup, and W = 8 workgroups that hold 2 KB of LDS (both workgroups of a persistent kernel still
        # fit beside one): what a second busy queue costs by itself, apart from the compute units it holds
        window = base * nsteps * 2.6 + 20.0
        for key, w, lds in (("one_workgroup_64KB", 1, 65536), ("8_workgroups_2KB", 8, 2048), ("64_workgroups_2KB", 64, 2048)):
            t = sorted(timed(step, nsteps, w, window, lds) for _ in range(3))[1]
            leg.setdefault("controls", {})[key] = {"ms_per_step": round(t, 3), "slowdown": round(t / base, 4)}
            print(name, key, leg["controls"][key], flush=True)
        res["legs"][name] = leg
        del step
        torch.cuda.empty_cache()
    s = json.dumps(res, indent=1)
    if out_path:
        open(out_path, "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
