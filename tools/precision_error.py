#!/usr/bin/env python3
"""Error of one big convolution (fwd, dgrad, wgrad) against fp64 in every precision mode: max |err| / max |ref|.
fp32 appears twice: "fp32" is the default path (Winograd F(2x2,3x3) on the fp32 MFMA when the layer is large enough),
"fp32_direct" the direct implicit GEMM (HPRI_WINOGRAD=0)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import torch
import torch.nn.functional as F
from hyperpri_amd import engine as E
from hyperpri_amd.autograd import run

dev = "cuda:0"
out = {}
for (N, Cin, H, W, Cout, ks) in [(1, 238, 304, 484, 64, 3), (2, 512, 76, 121, 512, 3), (1, 128, 152, 242, 128, 3), (1, 238, 76, 121, 64, 3)]:
    g = torch.Generator().manual_seed(7)
    x = torch.rand(N, Cin, H, W, generator=g)                      # reflectance-like: positive, mean 0.5
    w = (torch.rand(Cout, Cin, ks, ks, generator=g) * 2 - 1) / (Cin * ks * ks) ** 0.5
    r = torch.randn(N, Cout, H, W, generator=g)
    xc, wc = x.clone().double().requires_grad_(True), w.clone().double().requires_grad_(True)
    yc = F.conv2d(xc, wc, None, padding=ks // 2)
    yc.backward(r.double())
    row = {}
    for prec in ("fp32", "fp32_direct", "bf16x6", "bf16x3", "bf16"):
        E.WINOGRAD = prec != "fp32_direct"
        E.bump_param_epoch()
        mode = "fp32" if prec == "fp32_direct" else prec
        xd, wd = (t.to(dev).requires_grad_(True) for t in (x, w))
        yd = run(lambda tape, a, need: E.conv_bn_relu(tape, a[0], wd, None, None, True, ks, need_dx=need[0], precision=mode), [xd], [wd])
        yd.backward(r.to(dev))
        row[prec] = [float((yd.detach().cpu().double() - yc.detach()).abs().max() / yc.detach().abs().max()),
                     float((xd.grad.cpu().double() - xc.grad).abs().max() / xc.grad.abs().max()),
                     float((wd.grad.cpu().double() - wc.grad).abs().max() / wc.grad.abs().max())]
    row["fp32_path"] = "winograd" if E._wino_ok(E.Act.new(N, H, W, Cin, torch.device(dev)), Cout) else "direct (layer below the Winograd size threshold)"
    E.WINOGRAD = True
    out[f"{Cin}->{Cout} {H}x{W} batch {N}"] = row
    print(f"{Cin}->{Cout} {H}x{W}: " + "  ".join(f"{k}: fwd {v[0]:.1e} dgrad {v[1]:.1e} wgrad {v[2]:.1e}" for k, v in row.items() if isinstance(v, list)), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r02_precision_error.json", "w"), indent=1)
