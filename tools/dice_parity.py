#!/usr/bin/env python3
"""Dice/IoU parity on the emulated held-out split: fold 1 has 45 train / 14 val images
(train1.json / val1.json); no data ships with the reference, so cubes n = 45..58 are generator cubes
(seed 1234+n) with root-like polyline masks (seed 4321+n).  GPU-kernel logits vs CPU-oracle logits at identical
weights for (i) generator-init weights, train-mode BN, (ii) the same weights in eval mode after one train-mode pass
has populated the running statistics, (iii) eval mode after K Adam steps run by the HIP path.
Writes profiles/<tag>_dice_parity.json.   usage: dice_parity.py [n_val=14] [adam_steps=20] [tag=r01]"""
import json
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine, synth  # noqa: E402
from oracle import hyperpri_oracle as O  # noqa: E402

H, W, D = 608, 968, 238


def cube(n, dev):
    x = torch.empty((1, 1, D, H, W), device=dev)
    engine.synth_fill_(x[0], 1234 + n)
    return x


def compare(net, sd, val, train_mode, dev, rows, label):
    net.train(train_mode)
    for n in val:
        x = cube(n, dev)
        mask = torch.from_numpy(synth.polyline_mask(1, H, W, seed0=4321 + n))
        with torch.no_grad():
            lg = net(x).cpu()
            work = OrderedDict((k, v.clone()) for k, v in sd.items())
            lo = O.cubenet_forward(work, x.cpu(), 64, train_mode)
        a, d, i = O.seg_metrics(lg, mask)
        a2, d2, i2 = O.seg_metrics(lo, mask)
        flips = int(((lg > 0) != (lo > 0)).sum())
        rows.append({"variant": label, "cube": n, "max_abs_dlogit": float((lg - lo).abs().max()), "sign_flips": flips,
                     "dice_hip": d, "dice_oracle": d2, "iou_hip": i, "iou_oracle": i2, "acc_hip": a, "acc_oracle": a2,
                     "dice_equal_4dp": round(d, 4) == round(d2, 4), "iou_equal_4dp": round(i, 4) == round(i2, 4)})
        print(label, n, rows[-1]["max_abs_dlogit"], flips, d, d2, i, i2, flush=True)


def main():
    nval = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    ksteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    tag = sys.argv[3] if len(sys.argv) > 3 else "r01"
    precision = sys.argv[4] if len(sys.argv) > 4 else "fp32"
    if "--grads-fp32" in sys.argv:          # A/B: the activation gradients of the bf16 mode stored as fp32 again (round 3's storage)
        for a in ("SKIP_GRAD_BF16", "GRAD_BF16_SINGLE", "GRAD_BF16_INNER", "GRAD_BF16_GEMM"):
            setattr(engine, a, False)
    dev = torch.device("cuda", 0)
    torch.set_num_threads(bench.host_cores())
    val = list(range(45, 45 + nval))
    net = HP.CubeNET(D, 1, first_depth=64, bilinear=False)
    sd = O.synth_state_dict(O.cubenet_shapes(D, 1, 64))
    net.load_state_dict(sd)
    net = HP.set_precision(net.to(dev), precision)
    rows = []
    t0 = time.time()
    compare(net, sd, val, True, dev, rows, "init/train-BN")
    # (the train-mode forwards above advanced the HIP modules' running statistics 14 times; the oracle worked on
    #  copies) -> reset, then one train-mode pass over train cube 0 on both sides populates the running statistics
    net.load_state_dict(sd)
    net.train()
    x0 = cube(0, dev)
    with torch.no_grad():
        net(x0)
        O.cubenet_forward(sd, x0.cpu(), 64, True)
    compare(net, sd, val, False, dev, rows, "init/eval-BN")
    # K Adam steps on the HIP path (batch 2, lr 1e-3, BCEWithLogits: README.md:59), then identical weights on the CPU
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    crit = torch.nn.BCEWithLogitsLoss()
    losses = []
    for k in range(ksteps):
        n0 = (2 * k) % 44
        x = torch.cat([cube(n0, dev), cube(n0 + 1, dev)], 0)
        m = torch.from_numpy(synth.polyline_mask(2, H, W, seed0=4321 + n0)).to(dev)
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x), m)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    sd2 = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
    compare(net, sd2, val, False, dev, rows, f"adam{ksteps}/eval-BN")
    out = {"precision": precision, "cubes": val, "adam_steps": ksteps, "train_losses": losses, "rows": rows,
           "all_dice_equal_4dp": all(r["dice_equal_4dp"] for r in rows), "all_iou_equal_4dp": all(r["iou_equal_4dp"] for r in rows),
           "max_abs_dlogit": max(r["max_abs_dlogit"] for r in rows),
           "max_abs_ddice": max(abs(r["dice_hip"] - r["dice_oracle"]) for r in rows),
           "max_abs_diou": max(abs(r["iou_hip"] - r["iou_oracle"]) for r in rows),
           "max_sign_flip_fraction": max(r["sign_flips"] for r in rows) / float(H * W), "seconds": time.time() - t0}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for d in ("profiles", "gpurun_out"):
        with open(os.path.join(ROOT, d, f"{tag}_dice_parity.json"), "w") as f:
            json.dump(out, f, indent=1)
    print("summary:", {k: out[k] for k in ("precision", "all_dice_equal_4dp", "all_iou_equal_4dp", "max_abs_dlogit",
                                           "max_abs_ddice", "max_abs_diou", "max_sign_flip_fraction", "seconds")})


if __name__ == "__main__":
    main()
