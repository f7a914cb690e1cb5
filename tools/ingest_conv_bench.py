#!/usr/bin/env python3
"""hpri_conv3x3_ingest_h16 alone on the benched first layer (2 x 238x608x968 -> 64): HIP-event time over back-to-back launches after 1 s of
the same launches, beside layout pass + hpri_conv_bf16v3 (the pair it replaces).  usage: ingest_conv_bench.py [kind]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hyperpri_amd import _lib, engine as E  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda", 0)
N, C, H, W, F = 2, 238, 608, 968, 64
x = torch.empty((N, C, H, W), device=dev)
for i in range(N):
    E.synth_fill_(x[i], 1234 + i)
w = (torch.rand(F, C, 3, 3, device=dev) - 0.5) * 0.05
b = torch.zeros(F, device=dev)
bn = torch.nn.BatchNorm2d(F).to(dev).eval()
ref = E.BNRef(bn)


def timed(fn, settle=1.0, reps=50):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with _lib.using(None if kind == "bf16" else "f16"):
    fused = timed(lambda: E._conv_ingest_eval(E.Act.raw_nchw(x, -1), w, b, ref, C, F, True))
    pair = timed(lambda: E._conv_folded_eval(E.Act.from_tensor(x, -1), w, b, ref, 3, C, F, True, prec="bf16", inner=F))
    xa = E.Act.from_tensor(x, -1)
    conv_only = timed(lambda: E._conv_folded_eval(xa, w, b, ref, 3, C, F, True, prec="bf16", inner=F))
fl = 2.0 * N * H * W * 9 * C * F
print(json.dumps({"kind": kind, "stagger": os.environ.get("HPRI_IG_STAGGER"), "fused_ms": round(fused, 4), "fused_TF": round(fl / fused / 1e9, 1),
                  "pair_ms": round(pair, 4), "conv_only_ms": round(conv_only, 4), "layout_ms": round(pair - conv_only, 4),
                  "fused_gb_s_of_input": round(N * C * H * W * 4 / fused / 1e6, 1)}))
