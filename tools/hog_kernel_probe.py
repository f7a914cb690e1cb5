#!/usr/bin/env python3
"""Which kernels of the fp32 step lose time beside ONE resident hog workgroup (tools/cu_hog.hip, 64 KB of LDS)?  The step's
per-kernel HIP-event table (engine.enable_event_log) with and without the hog.  usage: hog_kernel_probe.py [precision] [wgrad_cu_reserve]"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
if len(sys.argv) > 2:
    engine.set_plan_option("wgrad_cu_reserve", int(sys.argv[2]))       # the fp32 Winograd weight gradient plans for that many fewer CUs
hog = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libcuhog.so"))
hog.cu_hog_launch_lds.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda", 0)
hs = torch.cuda.Stream(device=dev)
sink = torch.zeros(1024, dtype=torch.int32, device=dev)
net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
bench.synth_init_(net)
HP.set_precision(net, prec)
x = torch.empty((2, 1, 238, 608, 968), device=dev)
m = torch.empty((2, 1, 608, 968), device=dev)
for i in range(2):
    engine.synth_fill_(x[i], 1234 + i)
    engine.synth_fill_(m[i], 4321 + i, mode=1, thr=0.9)
crit = torch.nn.BCEWithLogitsLoss()


def step():
    for p in net.parameters():
        p.grad = None
    crit(net(x), m).backward()


for _ in range(4):
    step()
torch.cuda.synchronize()
out = {}
for side in (True, False):
    engine.SIDE_STREAM = side
    for label, w in (("alone", 0), ("one_hog_workgroup", 1)):
        step(); torch.cuda.synchronize()
        if w:
            assert hog.cu_hog_launch_lds(w, 400.0, 65536, sink.data_ptr(), hs.cuda_stream) == 0
            time.sleep(0.002)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            step()
        e1.record()
        torch.cuda.synchronize()
        whole = e0.elapsed_time(e1) / 3
        time.sleep(0.5)                      # (the hog has left)
        if w:
            assert hog.cu_hog_launch_lds(w, 400.0, 65536, sink.data_ptr(), hs.cuda_stream) == 0
            time.sleep(0.002)
        engine.enable_event_log(True)
        step()
        torch.cuda.synchronize()
        rows = engine.event_log_summary()
        engine.enable_event_log(False)
        time.sleep(0.5)
        out[f"side_stream={int(side)}/{label}"] = {"ms_per_step": round(whole, 3), "kernels_ms_per_step": {k: round(v["total_ms"], 3) for k, v in sorted(rows.items())}}
print(json.dumps(out, indent=1))
