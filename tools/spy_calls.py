#!/usr/bin/env python3
"""Which entry points one bf16 step of the benched workload calls, with the flag arguments of a few of them (diagnostic)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine as E  # noqa: E402

dev = torch.device("cuda", 0)
net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
bench.synth_init_(net)
HP.set_precision(net, sys.argv[1] if len(sys.argv) > 1 else "bf16")
x = torch.empty((2, 1, 238, 608, 968), device=dev)
E.synth_fill_(x, 1234)
mask = E.synth_fill_(torch.empty((2, 1, 608, 968), device=dev), 4321, mode=1, thr=0.9)
calls = []
real = E._lib.call


def spy(name, *a):
    calls.append((name, a))
    return real(name, *a)


torch.nn.BCEWithLogitsLoss()(net(x), mask).backward()
E._lib.call = spy
torch.nn.BCEWithLogitsLoss()(net(x), mask).backward()
E._lib.call = real
torch.cuda.synchronize()
cnt = collections.Counter(n for n, _ in calls)
for n, k in sorted(cnt.items()):
    print(f"{k:4d}  {n}")
print("y2 flags:", [a[-2] for n, a in calls if n == "hpri_conv_bf16v3_y2"])
print("pool bwd (x_bf16, dx_bf16):", [(a[1], a[8]) for n, a in calls if n == "hpri_maxpool2_bwd_x16"])
