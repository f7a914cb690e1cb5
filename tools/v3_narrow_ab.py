#!/usr/bin/env python3
"""conv_bf16v3 with / without the one column of 32 x 8 tiles for W mod 16 <= 8 (plan option bf16v3_tile_width 0 / 1): the 238->64 layer and
the whole bf16 / f16 step of the benched workload, arms interleaved.  usage: v3_narrow_ab.py > profiles/r05_v3_narrow_ab.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
import first_conv as FC  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
out = {"what": "plan option bf16v3_tile_width: 0 = one 32x8 tile column for a remainder of <= 8 columns (W = 968: 4598 items), 1 = 16-wide (4636 items)",
       "library_stamp": bench._lib_stamp(), "layer_238to64": {}, "step": {}}
for r in range(3):
    for opt in (1, 0):
        engine.set_plan_option("bf16v3_tile_width", opt)
        m = FC.measure(reps=10, settle_s=0.5, modes=("bf16_planes",))
        t = m.get("bf16_planes_bf16_out") or m["bf16_planes"]
        out["layer_238to64"].setdefault(str(opt), []).append({"ms": t["ms"], "TF": t["tflops"]})
net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
bench.synth_init_(net)
x = torch.empty((2, 1, 238, 608, 968), device=dev)
mask = torch.empty((2, 1, 608, 968), device=dev)
for i in range(2):
    engine.synth_fill_(x[i], 1234 + i)
    engine.synth_fill_(mask[i], 4321 + i, mode=1, thr=0.9)
crit = HP.BCEWithLogitsLoss()


def step():
    for p in net.parameters():
        p.grad = None
    loss = crit(net(x), mask)
    loss.backward()
    return loss


for prec in ("bf16", "f16"):
    HP.set_precision(net, prec)
    res = {"0": [], "1": []}
    for r in range(4):
        for opt in (1, 0):
            engine.set_plan_option("bf16v3_tile_width", opt)
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                l = step()
            torch.cuda.synchronize()
            res[str(opt)].append(round((time.perf_counter() - t0) / 30 * 1e3, 3))
    out["step"][prec] = {"ms": res, "median": {k: sorted(v)[len(v) // 2] for k, v in res.items()}, "loss": float(l)}
engine.set_plan_option("bf16v3_tile_width", 0)
print(json.dumps(out, indent=1))
