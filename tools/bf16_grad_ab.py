#!/usr/bin/env python3
"""Where the bf16 mode's gradient error comes from: the benched step's gradients against the reference's fp64
samples (tests/golden/grads_cubenet64_full_b2.npz) with the bf16 STORAGE of activation gradients switched off and on -- the MFMA
operands are bf16 planes in both arms.  arms: "grads_fp32" = SKIP_GRAD_BF16 / GRAD_BF16_SINGLE / GRAD_BF16_INNER / GRAD_BF16_GEMM
off (round 3's storage), "default" = all on (round 4).  usage: bf16_grad_ab.py > profiles/r05_bf16_grad_storage_ab.json"""
import json
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hyperpri_amd as H  # noqa: E402
from hyperpri_amd import engine  # noqa: E402
from oracle import hyperpri_oracle as O  # noqa: E402  (inputs and the fixture's sample positions only)
import test_gpu_deep_grads as T  # noqa: E402

DEV = "cuda:0"
z = np.load(os.path.join(ROOT, "tests", "golden", "grads_cubenet64_full_b2.npz"))
x = torch.cat([T._u(1234 + n, (1, 1, 238, 608, 968)) for n in range(2)], 0).to(DEV)
mask = torch.cat([(T._u(4321 + n, (1, 1, 608, 968)) > 0.9).float() for n in range(2)], 0).to(DEV)
ATTRS = ("SKIP_GRAD_BF16", "GRAD_BF16_SINGLE", "GRAD_BF16_INNER", "GRAD_BF16_GEMM")
out = {"fixture": "tests/golden/grads_cubenet64_full_b2.npz", "arms": {}}
for arm in ("grads_fp32", "default", "grads_fp32", "default"):
    saved = {a: getattr(engine, a) for a in ATTRS}
    if arm == "grads_fp32":
        for a in ATTRS:
            setattr(engine, a, False)
    try:
        net = H.CubeNET(238, 1, first_depth=64, bilinear=False)
        shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
        net.load_state_dict(O.synth_state_dict(shapes))
        net = H.set_precision(net.to(DEV), "bf16").train()
        loss = torch.nn.BCEWithLogitsLoss()(net(x), mask)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        for a, v in saved.items():
            setattr(engine, a, v)
    ns = int(z["ns"])
    rels, coss = {}, {}
    for k, (nm, p) in enumerate(net.named_parameters()):
        if p.dim() < 2 or float(z["grad_l2_64"][k]) < 1e-6:
            continue
        g = p.grad.detach().reshape(-1)
        idx = torch.from_numpy(T.sample_index(k, g.numel(), ns)).to(g.device)
        cnt = int(z["grad_sample_count"][k])
        hip, g64 = g[idx].double().cpu().numpy(), z["grad_sample64"][k, :cnt]
        rels[nm] = float(np.linalg.norm(hip - g64) / np.linalg.norm(g64))
        coss[nm] = float(np.dot(hip, g64) / (np.linalg.norm(hip) * np.linalg.norm(g64)))
    row = {"loss": float(loss.detach()), "worst_rel_l2": max(rels.values()), "median_rel_l2": float(np.median(list(rels.values()))),
           "worst_cosine": min(coss.values()), "per_stage_rel_l2": {k: round(v, 4) for k, v in rels.items() if k.endswith("double_conv.0.weight") or k.startswith(("first", "outc"))}}
    out["arms"].setdefault(arm, []).append(row)
    del net
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
