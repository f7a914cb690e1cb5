#!/usr/bin/env python3
"""Predict-path throughput (eval mode, no_grad) of CubeNET-64 on 238x608x968 cubes, with and without BN folding."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hyperpri_amd as HP
from hyperpri_amd import engine
dev = torch.device("cuda", 0)
net = HP.CubeNET(238, 1, 64, bilinear=False).to(dev)
bench.synth_init_(net)
x = torch.empty((2, 1, 238, 608, 968), device=dev)
engine.synth_fill_(x, 1234)
net.train()
with torch.no_grad():
    net(x)
net.eval()
for prec in ("fp32", "bf16"):
    HP.set_precision(net, prec)
    for fold in (False, True):
        engine.FOLD_EVAL_BN = fold
        with torch.no_grad():
            for _ in range(3):
                net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                net(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"eval forward {prec} fold={fold}: {dt * 1e3:.2f} ms per batch of 2 -> {2 / dt:.1f} cubes/s", flush=True)
