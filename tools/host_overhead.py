#!/usr/bin/env python3
"""Host (Python + ctypes + HIP launch) time of one training step against its GPU time: how far the launch thread runs ahead
of the device.  usage: host_overhead.py [precision]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda", 0)
net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
HP.set_precision(net, prec)
bench.synth_init_(net)
x = torch.empty((2, 1, 238, 608, 968), device=dev); engine.synth_fill_(x, 1234)
mask = engine.synth_fill_(torch.empty((2, 1, 608, 968), device=dev), 4321, mode=1, thr=0.9)
crit = HP.BCEWithLogitsLoss()


def step():
    for p in net.parameters():
        p.grad = None
    crit(net(x), mask).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"{prec}: host enqueue {host[len(host) // 2]:.2f} ms/step (min {host[0]:.2f}, max {host[-1]:.2f}); step wall {total[len(total) // 2]:.2f} ms; "
      f"cores {bench.host_cores()}, load {os.getloadavg()}")
