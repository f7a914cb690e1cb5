"""Full-size golden fixtures for BASELINE configs C5 and C3, from the REAL reference modules (build container only):

  net_cubenet128_300_full   CubeNET(300, 1, first_depth=128, bilinear=False) @ (1,1,300,608,968)   [config C5]
  net_spectral1650_full     SpectralUNET(238, 1, 1650)                       @ (1,238,608,700)      [config C3]

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_full2.py [c5] [c3]

Each fixture stores a sub-sample of the logits (train and eval mode), the BCE loss, logits mean/std, Acc/Dice/IoU,
the BatchNorm buffers after the step, and per-parameter gradient L2 norms plus the first 16 gradient values.

C3 note: stock autograd keeps ~88 GiB of activations for one 608x700 image, more than this
container has.  The reference's own sub-modules (``mod.tail``, ``mod.down1`` ... ``mod.outc``) are therefore driven in
the order of ``SpectralUNET.forward`` (models.py:129-145) with ``torch.utils.checkpoint`` around each of them: the same
ATen ops on the same tensors, only re-computed instead of stored.  The driver is checked against a plain ``mod(x)`` call
at a small size before the full-size run (``_check_driver``).
"""
import os
import sys

import numpy as np
import torch
from torch.utils.checkpoint import checkpoint

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as MG  # noqa: E402  (sets sys.path for the oracle and the reference)

O = MG.O


def spectral_ckpt_forward(mod, x):
    """SpectralUNET.forward (models.py:117-145) with every Linear-BN-ReLU module (and the concat in front of it)
    inside a checkpoint segment."""
    N, D, R, C = x.shape
    rast = x.reshape(N, D, R * C).permute(0, 2, 1)
    outs = []

    def seg(m):
        return lambda *t: m(t[0] if len(t) == 1 else torch.cat(t, -1))

    for in_x in rast:
        in_x = in_x.contiguous().requires_grad_(True)    # checkpoint needs one input that requires grad
        x0 = checkpoint(seg(mod.tail), in_x, use_reentrant=False)
        x1 = checkpoint(seg(mod.down1), x0, use_reentrant=False)
        x2 = checkpoint(seg(mod.down2), x1, use_reentrant=False)
        x3 = checkpoint(seg(mod.down3), x2, use_reentrant=False)
        x4 = checkpoint(seg(mod.down4), x3, use_reentrant=False)
        t = checkpoint(seg(mod.up1), x4, use_reentrant=False)
        t = checkpoint(seg(mod.up2), x3, t, use_reentrant=False)
        t = checkpoint(seg(mod.up3), x2, t, use_reentrant=False)
        t = checkpoint(seg(mod.up4), x1, t, use_reentrant=False)
        t = checkpoint(seg(mod.outc), x0, t, use_reentrant=False)
        outs.append(t.reshape(1, mod.n_classes, R, C))
    return torch.cat(outs, 0)


def _check_driver():
    torch.manual_seed(0)
    mod = MG.RM.SpectralUNET(22, 1, 50)
    MG.load_synth(mod, 1000, False)
    x = MG.u(1242, (2, 22, 9, 14))
    m = (MG.u(4324, (2, 1, 9, 14)) > 0.7).float()
    mod.train()
    a = mod(x)
    torch.nn.BCEWithLogitsLoss()(a, m).backward()
    g0 = [p.grad.clone() for p in mod.parameters()]
    for p in mod.parameters():
        p.grad = None
    b = spectral_ckpt_forward(mod, x)
    torch.nn.BCEWithLogitsLoss()(b, m).backward()
    assert torch.equal(a, b)
    for u, p in zip(g0, mod.parameters()):
        assert torch.allclose(u, p.grad, rtol=0, atol=0), "checkpointed driver differs from mod(x)"
    print("driver check ok")


def spectral_full():
    _check_driver()
    H, W = 608, 700
    mod = MG.RM.SpectralUNET(238, 1, 1650)
    MG.load_synth(mod, 1000, False)
    x = MG.u(1234, (1, 238, H, W))
    mask = (MG.u(4321, (1, 1, H, W)) > 0.9).float()
    stride = 97
    mod.train()
    logits = spectral_ckpt_forward(mod, x)
    bufs = MG.buffers_of(mod)                 # before backward: the checkpoint re-computation updates them again
    loss = torch.nn.BCEWithLogitsLoss()(logits, mask)
    print("forward done, loss", loss.item(), flush=True)
    loss.backward()
    acc, dice, iou = O.seg_metrics(logits.detach(), mask)
    lg = logits.detach().numpy()
    rec = {"loss": np.float64(loss.item()), "mean": np.float64(logits.detach().double().mean().item()),
           "std": np.float64(logits.detach().double().std().item()),
           "acc": np.float64(acc), "dice": np.float64(dice), "iou": np.float64(iou),
           "logits_sub": lg.reshape(-1)[::stride].copy(), "stride": np.int64(stride)}
    for k, b in bufs.items():
        rec["buf/" + k] = b
    gs = MG.summarize_grads(mod)
    rec["grad_names"] = np.array(list(gs.keys()))
    rec["grad_l2"] = np.array([v["l2"] for v in gs.values()], dtype=np.float64)
    rec["grad_head"] = np.array([v["head"] + [0.0] * (16 - len(v["head"])) for v in gs.values()], dtype=np.float32)
    for p in mod.parameters():
        p.grad = None
    # eval mode with the running statistics of exactly ONE training forward
    sd = mod.state_dict()
    for k, b in bufs.items():
        sd[k].copy_(torch.from_numpy(b))
    mod.eval()
    with torch.no_grad():
        le = mod(x).numpy()
    rec["logits_eval_sub"] = le.reshape(-1)[::stride].copy()
    np.savez_compressed(os.path.join(HERE, "net_spectral1650_full.npz"), **rec)
    print("wrote net_spectral1650_full loss", loss.item(), "dice", dice, "iou", iou)


def cubenet128_full():
    H, W = 608, 968
    mk = (MG.u(4321, (1, 1, H, W)) > 0.9).float()
    x = MG.u(1234, (1, 1, 300, H, W))
    MG.net_fixture("net_cubenet128_300_full", MG.RM.CubeNET(300, 1, first_depth=128, bilinear=False), x, mk,
                   full_logits=False, stride=97)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["c5", "c3"]
    if "c5" in which:
        cubenet128_full()
    if "c3" in which:
        spectral_full()
