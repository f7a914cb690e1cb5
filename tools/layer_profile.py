#!/usr/bin/env python3
"""Per-layer MFMA kernel timing for the C2 workload (CubeNET-64, batch 2, 608x968x238): HIP events around
every conv_fwd / conv_wgrad launch, keyed by kernel instantiation AND problem shape.  Diagnostic only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = sys.argv[1] if len(sys.argv) > 1 else "cube64"
    if model == "cube64":
        net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
        x = torch.empty((2, 1, 238, 608, 968), device=dev)
    else:
        net = HP.UNet(3, 1, bilinear=False).to(dev).train()
        x = torch.empty((2, 3, 608, 968), device=dev)
    bench.synth_init_(net)
    engine.synth_fill_(x, 1234)
    mask = engine.synth_fill_(torch.empty((2, 1, 608, 968), device=dev), 4321, mode=1, thr=0.9)
    crit = torch.nn.BCEWithLogitsLoss()
    engine.SHAPE_TAGS = True

    def step():
        for p in net.parameters():
            p.grad = None
        crit(net(x), mask).backward()
    step(); step()
    torch.cuda.synchronize()
    engine.enable_event_log(True)
    reps = 3
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    summ = engine.event_log_summary()
    tot = 0.0
    print(f"{'kernel / shape':88s} {'n/step':>6s} {'avg ms':>8s} {'TF/s':>7s} {'ms/step':>8s}")
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
        print(f"{k:88s} {v['launches'] // reps:6d} {v['avg_ms']:8.3f} {v['tflops']:7.1f} {v['total_ms'] / reps:8.3f}")
        tot += v["total_ms"] / reps
    print("MFMA kernels total ms/step:", round(tot, 2))


if __name__ == "__main__":
    main()
