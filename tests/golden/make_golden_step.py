"""Golden fixture for the caller-side tail of a step (SURVEY.md 8c "C1 counterpart", 8f rank 2): the REAL reference
UNet driven through ``training_step`` semantics (PLTrainer.py:79-98) with ``nn.BCEWithLogitsLoss`` and
``torch.optim.Adam(lr=1e-3)`` (PLTrainer.py:171-174) for three steps on generator-defined weights and inputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_step.py      (build container only)

Stored: per-step loss and (acc, dice, iou); after the last step every parameter's L2 norm and first 16 values, and the
eval-mode logits.  Data only -- no reference source text.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle import hyperpri_oracle as O  # noqa: E402
from src.Experiments import models as RM  # noqa: E402  (reference)


def u(seed, shape):
    return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())


def run(name, mod, x, mask, make_opt, steps=3):
    shp = OrderedDict((k, tuple(v.shape)) for k, v in mod.state_dict().items())
    mod.load_state_dict(O.synth_state_dict(shp, seed0=1000, bn_random=False))
    mod.train()
    opt = make_opt(mod.parameters())
    crit = torch.nn.BCEWithLogitsLoss()
    losses, mets = [], []
    for _ in range(steps):
        opt.zero_grad()
        pred = mod(x)
        loss = crit(pred, mask)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        mets.append(O.seg_metrics(pred, mask))
    rec = {"loss": np.array(losses, dtype=np.float64), "metrics": np.array(mets, dtype=np.float64)}
    names, l2, head = [], [], []
    for k, p in mod.named_parameters():
        v = p.detach().double().flatten()
        names.append(k); l2.append(float(v.norm())); head.append([float(q) for q in v[:16].float()] + [0.0] * max(0, 16 - v.numel()))
    rec["param_names"] = np.array(names)
    rec["param_l2"] = np.array(l2, dtype=np.float64)
    rec["param_head"] = np.array(head, dtype=np.float32)
    mod.eval()
    with torch.no_grad():
        rec["logits_eval"] = mod(x).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name, "losses", losses)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    h, w = 36, 50
    m = (u(4321, (2, 1, h, w)) > 0.9).float()
    run("step_unet3_tiny_adam", RM.UNet(3, 1, bilinear=False), u(1234, (2, 3, h, w)), m,
        lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=0))
    run("step_cubenet64_tiny_sgd", RM.CubeNET(6, 1, first_depth=64, bilinear=False), u(1235, (2, 1, 6, h, w)), m,
        lambda ps: torch.optim.SGD(ps, lr=1e-2, momentum=0.9, weight_decay=1e-4))


if __name__ == "__main__":
    main()
