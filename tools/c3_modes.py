#!/usr/bin/env python3
"""BASELINE config C3 (SpectralUNET-1650 @608x700) at full size in the precision modes: max |logit - reference fixture|, loss,
Dice/IoU against the fixture's, and ms per fwd+bwd step (batch 1, as the fixture).   usage: c3_modes.py [out.json] [modes...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_gpu_nets as T
from oracle import hyperpri_oracle as O      # checker only (tools/ are not the product path)

out = sys.argv[1] if len(sys.argv) > 1 else None
modes = sys.argv[2:] or ["fp32", "bf16x3", "bf16"]
z = T._load("net_spectral1650_full")
stride = int(z["stride"])
res = {}
for m in modes:
    net, xd, mask, lg, loss = T._full_size_step("c3", m)
    sub = lg.reshape(-1)[::stride].numpy()
    acc, dice, iou = O.seg_metrics(lg, mask)
    crit = torch.nn.BCEWithLogitsLoss()
    md = mask.to(xd.device)
    def step():
        for p in net.parameters(): p.grad = None
        crit(net(xd), md).backward()
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); step(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    res[m] = {"max_abs_dlogit": float(np.abs(sub - z["logits_sub"]).max()), "loss": loss, "ref_loss": float(z["loss"]),
              "dice": dice, "ref_dice": float(z["dice"]), "iou": iou, "ref_iou": float(z["iou"]), "ms_per_step_batch1": round(ms, 1),
              "model_tflops": round(77150.9 / ms, 1)}
    print(m, res[m], flush=True)
    del net, xd, mask, lg
    torch.cuda.empty_cache()
if out:
    json.dump(res, open(out, "w"), indent=1)
