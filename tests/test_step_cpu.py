"""Caller-side tail of a step -- CPU half: the oracle against the step fixtures captured from
the real reference + torch.optim (tests/golden/make_golden_step.py), the host logic of hyperpri_amd.trainer, and the
argument checks of the new C-ABI entry points (no compute call without a GPU)."""
import ctypes
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hyperpri_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def _u(seed, shape):
    return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())


def _oracle_train(shapes, forward, x, mask, make_opt, steps, **kw):
    sd = O.synth_state_dict(shapes, seed0=1000, bn_random=False)
    work, leaves = OrderedDict(), OrderedDict()
    for k, v in sd.items():
        if O.is_param(k):
            leaves[k] = work[k] = v.clone().requires_grad_(True)
        elif k.startswith("inc.0."):
            work[k] = work["first_conv." + k.split(".")[-1]]
        else:
            work[k] = v.clone()
    opt = make_opt(list(leaves.values()))
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = O.bce_with_logits(forward(work, x, train=True, **kw), mask)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    with torch.no_grad():
        le = forward(work, x, train=False, **kw)
    return losses, leaves, le


@pytest.mark.parametrize("name", ["step_unet3_tiny_adam", "step_cubenet64_tiny_sgd"])
def test_oracle_reproduces_reference_training_steps(name):
    z = np.load(os.path.join(G, name + ".npz"))
    m = (_u(4321, (2, 1, 36, 50)) > 0.9).float()
    if "unet3" in name:
        losses, leaves, le = _oracle_train(O.unet_shapes(3, 1), O.unet_forward, _u(1234, (2, 3, 36, 50)), m,
                                           lambda ps: torch.optim.Adam(ps, lr=1e-3, weight_decay=0), 3)
    else:
        losses, leaves, le = _oracle_train(O.cubenet_shapes(6, 1, 64), O.cubenet_forward, _u(1235, (2, 1, 6, 36, 50)), m,
                                           lambda ps: torch.optim.SGD(ps, lr=1e-2, momentum=0.9, weight_decay=1e-4), 3,
                                           first_depth=64)
    np.testing.assert_allclose(losses, z["loss"], rtol=0, atol=2e-6)
    names = [str(s) for s in z["param_names"]]
    assert names == list(leaves.keys())
    for i, k in enumerate(names):
        v = leaves[k].detach().double().flatten()
        assert abs(float(v.norm()) - z["param_l2"][i]) <= 1e-5 * z["param_l2"][i] + 2e-5, k
    np.testing.assert_allclose(le.numpy(), z["logits_eval"], rtol=0, atol=2e-3)


def test_pr_curve_known_answer():
    # 6 pixels, thresholds {0, .25, .5, .75, 1}: worked by hand
    p = torch.tensor([0.0, 0.2, 0.5, 0.5, 0.9, 1.0])
    t = torch.tensor([0, 1, 0, 1, 1, 1])
    prec, rec, thr, tp, fp, fn = O.pr_curve_binned(p, t, thresholds=5)
    assert thr.tolist() == [0.0, 0.25, 0.5, 0.75, 1.0]
    assert tp.tolist() == [4, 3, 3, 2, 1] and fp.tolist() == [2, 1, 1, 0, 0] and fn.tolist() == [0, 1, 1, 2, 3]
    np.testing.assert_allclose(prec.numpy(), [4 / 6, 3 / 4, 3 / 4, 1, 1, 1])
    np.testing.assert_allclose(rec.numpy(), [1, 3 / 4, 3 / 4, 1 / 2, 1 / 4, 0])


def test_prcurve_host_logic_matches_oracle():
    from hyperpri_amd.trainer import PRCurve, best_dice_threshold
    T = 500
    p = _u(77, (40000,)) ** 2
    p[:7] = torch.tensor([0.0, 1.0, 0.5, 0.25, 1.0 / 499, 498.0 / 499, 0.1])   # values that sit on thresholds
    t = (_u(78, (40000,)) < p).float()
    thr = torch.linspace(0, 1, T)
    b = (p[:, None] >= thr[None, :]).sum(dim=1)                                # the bin the HIP kernel computes
    hist = torch.zeros(2, T + 1, dtype=torch.int64)
    for c in (0, 1):
        hist[c] = torch.bincount(b[t == c], minlength=T + 1)
    pr = PRCurve(T)
    pr.thresholds, pr.hist = thr, hist.flatten()
    prec, rec, th = pr.compute()
    oprec, orec, oth, otp, ofp, ofn = O.pr_curve_binned(p, t, T)
    tp, fp, fn, tn = pr.confusion()
    assert torch.equal(tp, otp) and torch.equal(fp, ofp) and torch.equal(fn, ofn)
    assert torch.equal(tp + fp + fn + tn, torch.full((T,), 40000))
    assert torch.equal(prec, oprec) and torch.equal(rec, orec) and torch.equal(th, oth)
    assert best_dice_threshold(prec, rec, th) == O.best_dice_threshold(oprec, orec, oth)


def test_metrics_from_counts_match_oracle_definitions():
    from hyperpri_amd.trainer import metrics_from_counts
    lg = _u(5, (2, 1, 20, 30)) * 4 - 2
    m = (_u(6, (2, 1, 20, 30)) > 0.6).float()
    tp, fp, fn, tn = O.seg_counts(lg, m, 0.5)
    acc, dice, iou = O.seg_metrics(lg, m, 0.5)
    got = metrics_from_counts(tp, fp, fn, tn)
    assert abs(got["acc"] - acc) < 1e-12 and abs(got["dice"] - dice) < 1e-12 and abs(got["pos_iou"] - iou) < 1e-12
    assert metrics_from_counts(0, 0, 0, 10)["dice"] == 1e-12          # zero_division=1e-12 (PLTrainer.py:66)


def test_checkpoint_key_translation():
    from hyperpri_amd.trainer import network_state_dict
    w = torch.zeros(1)
    want = ["inc.double_conv.0.weight", "outc.conv.bias"]
    lightning = {"pytorch-lightning_version": "2.0.7", "state_dict": OrderedDict(("m_network." + k, w) for k in want)}
    assert list(network_state_dict(lightning)) == want
    assert list(network_state_dict(OrderedDict((k, w) for k in want))) == want
    assert list(network_state_dict(OrderedDict(("module." + k, w) for k in want))) == want
    ds = OrderedDict(("_forward_module.m_network." + k, w) for k in want)
    ds["_forward_module.m_network.feat_ext.0.weight"] = w
    assert list(network_state_dict(ds)) == want


def test_step_entry_points_reject_bad_arguments():
    from hyperpri_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.hpri_bce_workspace_doubles(1) == 1 and lib.hpri_bce_workspace_doubles(10 ** 9) == 1024
    assert lib.hpri_bce_logits_fwd(null, null, 10, null, null, 0, null) == -1
    assert lib.hpri_bce_logits_bwd(null, null, 10, null, null, null) == -1
    assert lib.hpri_seg_counts(null, null, 10, 0.5, 1, null, null) == -1
    one = ctypes.c_void_p(16)      # non-null, never dereferenced: the checks fail first
    assert lib.hpri_pr_curve_hist(one, one, 10, one, 1, 0, one, null) == -1 and b"thresholds" in lib.hpri_last_error()
    assert lib.hpri_pr_curve_hist(one, one, 10, one, 5000, 0, one, null) == -1
    assert lib.hpri_adam_step(null, null, null, null, null, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, null, null) == -1
    arr = (ctypes.c_void_p * 1)(None)
    n = (ctypes.c_longlong * 1)(4)
    assert lib.hpri_adam_step(arr, arr, arr, arr, n, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, null, null) == -1
    assert lib.hpri_adam_step(arr, arr, arr, arr, n, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, null, null) == -1   # step >= 1
    assert lib.hpri_sgd_step(arr, arr, null, n, 1, 1e-2, 0.9, 0.0, 1, null, null) == -1                      # momentum w/o buffers


def test_trainer_fails_loudly_on_cpu_tensors():
    from hyperpri_amd.trainer import BCEWithLogitsLoss, FusedAdam, SegCounts
    with pytest.raises(RuntimeError):
        BCEWithLogitsLoss()(torch.zeros(4), torch.zeros(4))
    with pytest.raises(RuntimeError):
        SegCounts().update(torch.zeros(4), torch.zeros(4))
    p = torch.nn.Parameter(torch.zeros(4)); p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        FusedAdam([p]).step()


def test_ingest_entry_points_reject_bad_arguments():
    from hyperpri_amd import _lib
    from hyperpri_amd.ingest import CubeStager, from_hwb
    lib = _lib.load()
    null, one = ctypes.c_void_p(0), ctypes.c_void_p(16)
    assert lib.hpri_hwb_ingest(null, 0, null, 10, 8, 0, 8, 8, 8, null) == -1
    assert lib.hpri_hwb_ingest(one, 0, one, 10, 8, 4, 8, 8, 8, null) == -1 and b"band range" in lib.hpri_last_error()
    assert lib.hpri_hwb_ingest(one, 0, one, 10, 8, 0, 6, 6, 6, null) == -1      # destination stride not a multiple of 4
    assert lib.hpri_hwb_ingest(one, 2, one, 10, 8, 0, 8, 8, 8, null) == -1      # unknown dtype
    assert lib.hpri_hwb_h2d(null, null, 10, 8, 0, 8, 8, null) == -1
    with pytest.raises(RuntimeError):
        CubeStager(1, 4, 4, 8, device="cpu")
    with pytest.raises(RuntimeError):
        from_hwb(torch.zeros(1, 4, 4, 8))
