"""precision="f16": the plane paths of the bf16 mode with IEEE half as the 16-bit type (libhyperpri_hip_f16.so, the same sources built
with -DHPRI_H16_F16) -- the dtype BASELINE.json's north_star names for the MFMA roofline claim (fp16 operands
move the reference's logits by 3.5e-3 where bf16 operands move them by 2.3e-2; models.py:169, model_parts.py:22-27).  Activations,
pre-BN tensors and activation gradients are stored as half; the head multiplies the loss gradient by a power of two (the next one above
the number of logits) and the parameter gradients lose it again when they leave the tape.  Gates: against the REFERENCE's fixtures
(logits, loss, Dice / IoU, gradients against its fp64 samples), and against the bf16 mode on the same inputs (f16 must be the more
accurate of the two).  Needs a real MI355X: ``-m gpu``."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from conftest import record_margin
from oracle import hyperpri_oracle as O
import test_gpu_deep_grads as DG
import test_gpu_nets as TN

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _step(net, x, mask):
    for p in net.parameters():
        p.grad = None
    logits = net(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, mask)
    loss.backward()
    torch.cuda.synchronize()
    return logits.detach(), float(loss.detach()), [p.grad.detach().clone() for p in net.parameters()]


TINY = [c for c in TN.CASES if c[0] in ("net_unet3_tiny", "net_cubenet64_tiny", "net_cubenet128_tiny", "net_spectral_f50", "net_cubenet128_300_small",
                                        "net_spectral1650_small")]


@pytest.mark.parametrize("name,xseed,xshape,mseed,thr", TINY, ids=[c[0] for c in TINY])
def test_tiny_net_f16_mode(name, xseed, xshape, mseed, thr):
    """Small networks (incl. configs C3 / C5 at their exact channel widths): the f16 mode's logits are closer to the reference
    fixture than the bf16 mode's on the same inputs, the loss agrees, every gradient is finite and its norm is the fixture's."""
    import hyperpri_amd as H
    from hyperpri_amd import _lib
    z = TN._load(name)
    x = TN._u(xseed, xshape).to(DEV)
    mask = (TN._u(mseed, (xshape[0], int(z["logits"].shape[1])) + tuple(xshape[-2:])) > thr).float().to(DEV)
    errs = {}
    for prec in ("bf16", "f16"):
        net = TN._mk(name)
        shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
        net.load_state_dict(O.synth_state_dict(shapes))
        net = H.set_precision(net.to(DEV), prec).train()
        lg, loss, grads = _step(net, x, mask)
        assert _lib.kind() == "bf16"                      # (the library selection does not leak out of the tape)
        errs[prec] = float(np.abs(lg.cpu().numpy() - z["logits"]).max())
        assert abs(loss - float(z["loss"])) < (5e-3 if prec == "bf16" else 1e-3), (prec, loss)
        for (k, p), g in zip(net.named_parameters(), grads):
            assert torch.isfinite(g).all(), (prec, k)
        if prec == "f16":
            for i, k in enumerate(list(z["grad_names"])):
                g = dict(zip([n for n, _ in net.named_parameters()], grads))[k]
                ref = float(z["grad_l2"][i])
                if g.dim() > 1 and ref > 1e-6:
                    assert abs(float(g.double().norm()) - ref) <= 0.2 * ref, (k, float(g.double().norm()), ref)
    record_margin(f"f16/tiny/{name}/logits", errs["f16"], 2e-2)
    record_margin(f"f16/tiny/{name}/logits_over_bf16", errs["f16"] / max(errs["bf16"], 1e-9), 0.6)
    assert errs["f16"] < 2e-2 and errs["f16"] < 0.6 * errs["bf16"], errs


def test_full_size_c2_batch2_f16_vs_reference_fixture():
    """The benched workload in the f16 mode against the reference fixture (tests/golden/grads_cubenet64_full_b2.npz): logits within
    8e-3 (bf16: 5e-2), <= 0.1 % sign flips, loss within 2e-4, Dice / IoU within 1e-4, every gradient tensor against the fp64 samples
    (relative L2 and cosine per tensor, recorded in gpurun_out/f16_grad_parity_c2.json; bands = measured + 50 %)."""
    import hyperpri_amd as H
    z = np.load(os.path.join(G, "grads_cubenet64_full_b2.npz"))
    Hh, Ww = 608, 968
    net = H.CubeNET(238, 1, first_depth=64, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = H.set_precision(net.to(DEV), "f16").train()
    x = torch.cat([DG._u(1234 + n, (1, 1, 238, Hh, Ww)) for n in range(2)], 0)
    mask = torch.cat([(DG._u(4321 + n, (1, 1, Hh, Ww)) > 0.9).float() for n in range(2)], 0)
    lg, loss, _ = _step(net, x.to(DEV), mask.to(DEV))
    lgc = lg.cpu()
    stride = int(z["stride"])
    sub = lgc.reshape(-1)[::stride].numpy()
    d = float(np.abs(sub - z["logits_sub"]).max())
    flips = float(((sub > 0) != (z["logits_sub"] > 0)).mean())
    record_margin("f16/c2_batch2/logits", d, 8e-3)
    record_margin("f16/c2_batch2/sign_flips", flips, 1e-3)
    assert d <= 8e-3 and flips <= 1e-3, (d, flips)
    assert abs(loss - float(z["loss64"])) < 2e-4
    acc, dice, iou = O.seg_metrics(lgc, mask)
    assert abs(dice - float(z["dice"])) <= 1e-4 and abs(iou - float(z["iou"])) <= 1e-4
    ns = int(z["ns"])
    rows, worst_w, worst_cos, worst_norm = {}, 0.0, 1.0, 0.0
    for k, (nm, p) in enumerate(net.named_parameters()):
        g = p.grad.detach().reshape(-1)
        assert torch.isfinite(g).all(), nm
        idx = torch.from_numpy(DG.sample_index(k, g.numel(), ns)).to(g.device)
        cnt = int(z["grad_sample_count"][k])
        hip, g64 = g[idx].double().cpu().numpy(), z["grad_sample64"][k, :cnt]
        if float(z["grad_l2_64"][k]) < 1e-6:
            assert float(np.abs(hip).max()) <= 1e-4, nm
            continue
        rel = float(np.linalg.norm(hip - g64) / np.linalg.norm(g64))
        cos = float(np.dot(hip, g64) / (np.linalg.norm(hip) * np.linalg.norm(g64) + 1e-300))
        l2 = float(g.double().norm()) / float(z["grad_l2_64"][k])
        rows[nm] = {"rel_l2": rel, "cosine": cos, "norm_over_fp64_norm": l2}
        if p.dim() >= 2:
            worst_w, worst_norm = max(worst_w, rel), max(worst_norm, abs(l2 - 1.0))
        if cnt >= 16:
            worst_cos = min(worst_cos, cos)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "f16_grad_parity_c2.json"), "w") as f:
        json.dump({"what": "CubeNET-64, 2 x 238x608x968, precision f16: logits against the reference fixture, gradients against its fp64 samples",
                   "max_abs_dlogit": d, "sign_flip_fraction": flips, "loss": loss, "loss_fp64": float(z["loss64"]), "worst_rel_l2_weights": worst_w,
                   "worst_cosine": worst_cos, "worst_weight_norm_error": worst_norm, "tensors": rows}, f, indent=1)
    record_margin("f16/c2_batch2/grad_rel_l2_weights", worst_w, 0.32)
    record_margin("f16/c2_batch2/one_minus_cosine", 1.0 - worst_cos, 0.037)
    assert worst_w <= 0.32 and worst_cos >= 0.963 and worst_norm <= 0.01, (worst_w, worst_cos, worst_norm)      # measured 0.209 / 0.9757 / 0.0022 (bf16: 0.584 / 0.824 / 0.0054)


def test_f16_step_is_deterministic_and_leaves_the_bf16_library_alone():
    """Two f16 steps are bit-identical; a bf16 step after them equals a bf16 step before them (separate pack caches and queues)."""
    import hyperpri_amd as H
    net = H.CubeNET(6, 1, first_depth=64, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = net.to(DEV).train()
    x = TN._u(1235, (2, 1, 6, 64, 96)).to(DEV)
    mask = (TN._u(4321, (2, 1, 64, 96)) > 0.9).float().to(DEV)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    out = {}
    for tag, prec in (("b0", "bf16"), ("h0", "f16"), ("h1", "f16"), ("b1", "bf16")):
        net.load_state_dict(sd)
        H.set_precision(net, prec)
        out[tag] = _step(net, x, mask)
    for a, b in (("h0", "h1"), ("b0", "b1")):
        assert torch.equal(out[a][0], out[b][0]) and all(torch.equal(p, q) for p, q in zip(out[a][2], out[b][2])), (a, b)
    assert not torch.equal(out["h0"][0], out["b0"][0])


def test_full_size_spectral1650_f16_vs_reference_fixture():
    """BASELINE config C3 (SpectralUNET-1650 @608x700, batch 1) in the f16 mode against the reference fixture: the whole Linear stack
    on the plane GEMM / weight-gradient kernels of the half-precision library.  Logits (bf16 mode: 1.7e-2), loss, sign agreement,
    Dice / IoU, gradient norms and gradient heads."""
    z = TN._load("net_spectral1650_full")
    net, xd, mask, lg, loss = TN._full_size_step("c3", "f16")
    stride = int(z["stride"])
    sub = lg.reshape(-1)[::stride].numpy()
    d = float(np.abs(sub - z["logits_sub"]).max())
    record_margin("f16/c3/logits", d, 6e-3)
    assert d < 6e-3 and abs(loss - float(z["loss"])) < 5e-5, (d, loss)
    assert float(((sub > 0) != (z["logits_sub"] > 0)).mean()) < 2e-3
    acc, dice, iou = O.seg_metrics(lg, mask)
    assert abs(dice - float(z["dice"])) < 1e-4 and abs(iou - float(z["iou"])) < 1e-4
    grads = OrderedDict((k, p.grad) for k, p in net.named_parameters())
    worst = 0.0
    for i, k in enumerate(list(z["grad_names"])):
        if grads[k].dim() <= 1:
            continue
        assert torch.isfinite(grads[k]).all(), k
        g, ref = float(grads[k].detach().double().norm()), float(z["grad_l2"][i])
        worst = max(worst, abs(g - ref) / (ref + 1e-12))
    record_margin("f16/c3/grad_norms", worst, 0.05)
    assert worst <= 0.05, worst
    TN.check_heads_lowp(z, net, "f16/c3", 0.3)
