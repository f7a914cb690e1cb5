#!/usr/bin/env python3
"""Predict path (eval, inference_mode; PLTrainer.py:530-532) of CubeNET-64 on two 238x608x968 cubes: forward time in the 16-bit modes
with the first layer reading the caller's cube itself (csrc/conv_ingest.hip) against the layout pass + plane convolution it replaces,
arms interleaved; per-kernel HIP-event times of the first layer in both forms; logits compared bit for bit.
usage: predict_bench.py [rounds] > profiles/r05_predict_ingest.json"""
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
bench.synth_init_(net)
x = torch.empty((2, 1, 238, 608, 968), device=dev)
for i in range(2):
    engine.synth_fill_(x[i], 1234 + i)
with torch.no_grad():
    net(x)                      # running statistics of one training forward
net.eval()
out = {"what": "CubeNET-64 predict forward (eval, inference_mode), 2 x 238x608x968; fused = first layer reads the caller's NC(D)HW cube "
               "(hpri_conv3x3_ingest_h16), pair = layout pass + hpri_conv_bf16v3", "library_stamp": bench._lib_stamp(), "modes": {}}


def fwd():
    with torch.inference_mode():
        return net(x)


for prec in ("bf16", "f16", "fp32"):
    HP.set_precision(net, prec)
    times = {"fused": [], "pair": []}
    logits = {}
    for r in range(rounds):
        for arm in ("pair", "fused"):
            engine.INGEST_FUSED = arm == "fused"
            for _ in range(5):
                y = fwd()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                y = fwd()
            torch.cuda.synchronize()
            times[arm].append((time.perf_counter() - t0) / 30 * 1e3)
            logits[arm] = y.clone()
    kern = {}
    for arm in ("pair", "fused"):
        engine.INGEST_FUSED = arm == "fused"
        engine.SHAPE_TAGS = True
        engine.enable_event_log(True)
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        summ = engine.event_log_summary()
        engine.enable_event_log(False)
        engine.SHAPE_TAGS = False
        kern[arm] = {k: {"avg_ms": round(v["avg_ms"], 4), "tflops": round(v["tflops"], 1), "launches": v["launches"] // 3}
                     for k, v in summ.items() if "608x968" in k and ("C238" in k or "C256" in k)}
    engine.INGEST_FUSED = True
    engine.enable_event_log(True)            # the whole forward, per kernel family (MFMA kernels only: the others carry no tag)
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    whole = {k: {"ms_per_forward": round(v["total_ms"] / 3, 4), "launches": v["launches"] // 3, "tflops": round(v["tflops"], 1)}
             for k, v in sorted(engine.event_log_summary().items(), key=lambda kv: -kv[1]["total_ms"])}
    engine.enable_event_log(False)
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    out["modes"][prec] = {"ms": {k: [round(t, 3) for t in v] for k, v in times.items()}, "median_ms": {k: round(v, 3) for k, v in med.items()},
                          "fused_over_pair": round(med["fused"] / med["pair"], 4), "logits_bit_identical": bool(torch.equal(logits["fused"], logits["pair"])),
                          "ingest_launches_seen": engine.INGEST_LAUNCHES, "first_layer_kernels": kern, "mfma_kernels_of_the_fused_forward": whole}
print(json.dumps(out, indent=1))
