#!/usr/bin/env python3
"""Run one BASELINE.json config (C1..C5 shapes) for a few fwd+bwd steps on cuda:0 and print time, model TFLOP/s
and peak memory.  usage: run_config.py {unet3|cube64|cube128|spectral} [batch] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine  # noqa: E402

CFG = {
    "unet3": (lambda: HP.UNet(3, 1, bilinear=False), lambda b: (b, 3, 608, 968), 2591.51),
    "cube64": (lambda: HP.CubeNET(238, 1, 64, bilinear=False), lambda b: (b, 1, 238, 608, 968), 2910.17),
    "cube128": (lambda: HP.CubeNET(300, 1, 128, bilinear=False), lambda b: (b, 1, 300, 608, 968), 3986.84),
    "spectral": (lambda: HP.SpectralUNET(238, 1, 1650), lambda b: (b, 238, 608, 700), 77150.90),
}


def main():
    name = sys.argv[1]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    mk, shp, gflop = CFG[name]
    dev = torch.device("cuda", 0)
    net = mk().to(dev).train()
    bench.synth_init_(net)
    x = torch.empty(shp(batch), device=dev)
    for i in range(batch):
        engine.synth_fill_(x[i], 1234 + i)
    hw = x.shape[-2:]
    mask = torch.empty((batch, 1) + tuple(hw), device=dev)
    for i in range(batch):
        engine.synth_fill_(mask[i], 4321 + i, mode=1, thr=0.9)
    crit = torch.nn.BCEWithLogitsLoss()

    def step():
        for p in net.parameters():
            p.grad = None
        loss = crit(net(x), mask)
        loss.backward()
        return loss
    loss = step()
    torch.cuda.synchronize()
    if os.environ.get("EVENTS"):            # per-kernel HIP-event table (serialises launches a little)
        engine.SHAPE_TAGS = True
        engine.enable_event_log(True)
        step()
        torch.cuda.synchronize()
        rows = engine.event_log_summary()
        engine.enable_event_log(False)
        tot = 0.0
        for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["total_ms"]):
            tot += v["total_ms"]
            print(f"{k:90s} n={v['launches']:3d} {v['total_ms']:9.3f} ms {v['tflops']:7.1f} TF")
        print("sum of timed kernels", round(tot, 2), "ms")
    for _ in range(int(os.environ.get("WARM", "1")) - 1):      # further untimed steps (the allocator settles after two or three)
        step()
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms1 = torch.cuda.memory_stats()
    print("allocator in the timed steps: device mallocs", ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0), "frees",
          ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0), "retries", ms1.get("num_alloc_retries", 0) - ms0.get("num_alloc_retries", 0),
          "reserved GiB", round(torch.cuda.memory_reserved() / 2**30, 1), flush=True)
    print(f"{name} batch {batch}: {dt * 1e3:.1f} ms/step, {batch / dt:.3f} units/s, {batch * gflop / dt / 1e3:.1f} model TFLOP/s, "
          f"loss {float(loss.detach()):.6f}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
