#!/usr/bin/env python3
"""Predict forward (eval, inference_mode) of the BASELINE configurations' networks on one GPU: time per forward and the MFMA kernel families
it spends it in, per precision mode.  usage: predict_configs.py {unet3|cube64|cube128|spectral} [batch] > profiles/r05_predict_<name>.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine  # noqa: E402
from run_config import CFG  # noqa: E402

name = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mk, shp, gflop = CFG[name]
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
net = mk().to(dev).train()
bench.synth_init_(net)
x = torch.empty(shp(batch), device=dev)
for i in range(batch):
    engine.synth_fill_(x[i], 1234 + i)
with torch.no_grad():
    net(x)
net.eval()
out = {"what": f"{name} predict forward (eval, inference_mode), batch {batch}, input {tuple(x.shape)}", "library_stamp": bench._lib_stamp(), "modes": {}}
for prec in ("bf16", "f16", "fp32"):
    HP.set_precision(net, prec)
    with torch.inference_mode():
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        n = 10 if name == "spectral" else 30
        t0 = time.perf_counter()
        for _ in range(n):
            y = net(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        engine.enable_event_log(True)
        for _ in range(2):
            net(x)
        torch.cuda.synchronize()
        summ = engine.event_log_summary()
        engine.enable_event_log(False)
    out["modes"][prec] = {"ms_per_forward": round(ms, 3), "forward_model_tflops": round(batch * gflop / 3.0 / ms, 1),
                          "finite": bool(torch.isfinite(y).all()), "plane_conversions_so_far": engine.PLANE_CONVERSIONS,
                          "mfma_kernels": {k: {"ms_per_forward": round(v["total_ms"] / 2, 4), "launches": v["launches"] // 2, "tflops": round(v["tflops"], 1)}
                                           for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}}
print(json.dumps(out, indent=1))
