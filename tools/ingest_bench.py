#!/usr/bin/env python3
"""PCIe-inclusive throughput of the C2 workload (CubeNET-64, 238 bands, 608x968, batch 2, fwd+bwd) -- DESIGN.md 4.

Three ways to get a batch onto the device, each followed by the same training step:
  resident   the cube is already in HBM in the reference's (N,1,B,H,W) layout (what bench.py's `value` times)
  reference  what dataset.py + Lightning do: a pageable (N,1,B,H,W) host tensor, `.to(device)` inside the step
  stager     hyperpri_amd.ingest.CubeStager: pinned (N,H,W,B) slot, H2D of batch k+1 overlapped with step k,
             the copy lands directly in the network's channels-last layout (direct) or via one pad pass (kernel),
             fp32 or fp16 source
usage: python tools/ingest_bench.py [steps]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hyperpri_amd as H  # noqa: E402
from hyperpri_amd.ingest import CubeStager  # noqa: E402
from bench import synth_init_  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
N, B, HH, WW = 2, 238, 608, 968
net = H.CubeNET(B, 1, first_depth=64, bilinear=False).to(dev).train()
synth_init_(net)
crit = H.BCEWithLogitsLoss()
mask = (torch.rand(N, 1, HH, WW, device=dev) > 0.9).float()


def step(x):
    for p in net.parameters():
        p.grad = None
    loss = crit(net(x), mask)
    loss.backward()
    return loss


def timed(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(n):
        fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


out = {}
x_res = torch.rand(N, 1, B, HH, WW, device=dev)
for _ in range(2):
    step(x_res)
out["resident_ms"] = timed(lambda k: step(x_res), STEPS) * 1e3

x_host = torch.rand(N, 1, B, HH, WW)                       # pageable, as the default collate produces
step(x_host.to(dev))
out["reference_pageable_ms"] = timed(lambda k: step(x_host.to(dev)), max(2, STEPS // 2)) * 1e3
x_pin = x_host.pin_memory()
out["reference_pinned_sync_ms"] = timed(lambda k: step(x_pin.to(dev, non_blocking=True)), max(2, STEPS // 2)) * 1e3
del x_host, x_pin, x_res

for name, kw in (("stager_direct_f32", dict()), ("stager_kernel_f32", dict(direct_h2d=False)),
                 ("stager_kernel_f16", dict(src_dtype=np.float16))):
    st = CubeStager(N, HH, WW, B, device=dev, **kw)
    for s in range(st.slots):
        a = st.host_slot(s)
        a[...] = np.random.default_rng(s).random(a.shape, dtype=np.float32).astype(a.dtype)
    # H2D alone
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4):
        st.submit(); st.release()
    torch.cuda.synchronize()
    out[name + "_h2d_only_ms"] = (time.perf_counter() - t) / 4 * 1e3
    # pipelined: submit batch k+1 before running step k
    x = st.submit()

    def piped(k):
        global x
        cur = x
        loss = step(cur)
        st.release()
        x = st.submit()          # overlaps with the step just enqueued (side stream)
        return loss
    piped(0)
    out[name + "_pipelined_ms"] = timed(piped, STEPS) * 1e3
    del st, x
    torch.cuda.empty_cache()

for k in list(out):
    if k.endswith("_ms") and "h2d_only" not in k:
        out[k.replace("_ms", "_cubes_per_s")] = round(N / out[k] * 1e3, 2)
out = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()}
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r01_ingest.json", "w"), indent=1)
