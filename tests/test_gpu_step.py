"""Caller-side tail of a step on the HIP path (csrc/step.hip through hyperpri_amd.trainer) vs the CPU oracle, torch's
own CPU implementations of BCEWithLogitsLoss / Adam / SGD, and the step fixtures captured from the real reference.
Tolerances: loss 1e-6 relative; counts and histograms bit-exact on identical probabilities; optimizer updates within
a few ulp of the parameter scale (rtol 2e-6, atol 5e-8 at |p| <= 0.1).  Needs a real MI355X: ``-m gpu``."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hyperpri_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _u(seed, shape):
    return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())


@pytest.mark.parametrize("shape", [(1, 1, 3, 5), (2, 1, 36, 50), (2, 1, 608, 968)])
def test_bce_with_logits_matches_torch_cpu(shape):
    from hyperpri_amd.trainer import BCEWithLogitsLoss
    x = (_u(31, shape) * 16 - 8)
    x.view(-1)[:3] = torch.tensor([0.0, 40.0, -40.0])[: min(3, x.numel())]
    y = (_u(32, shape) > 0.9).float()
    xr = x.double().requires_grad_(True)
    ref = torch.nn.BCEWithLogitsLoss()(xr, y.double())
    ref.backward()
    xg = x.to(DEV).requires_grad_(True)
    loss = BCEWithLogitsLoss()(xg, y.to(DEV))
    assert loss.shape == () and loss.dtype == torch.float32
    (loss * 3.0).backward()                       # non-unit upstream gradient
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-6 * abs(float(ref.detach()))
    np.testing.assert_allclose(xg.grad.cpu().double().numpy(), 3.0 * xr.grad.numpy(), rtol=2e-6, atol=1e-12)
    # fixed-order reduction: bitwise reproducible
    again = BCEWithLogitsLoss()(xg.detach(), y.to(DEV))
    assert torch.equal(again, loss.detach())


def test_seg_counts_exact():
    from hyperpri_amd.trainer import SegCounts
    lg = _u(41, (2, 1, 608, 968)) * 6 - 3
    m = (_u(42, (2, 1, 608, 968)) > 0.85).float()
    for thr in (0.5, 0.31):
        c = SegCounts(thr)
        c.update(lg.to(DEV), m.to(DEV))
        c.update(lg.to(DEV), m.to(DEV).to(torch.int32))          # accumulates; integer masks accepted
        got = c.compute()
        # decide on the SAME fp32 probabilities: bit-exact integers
        p = torch.sigmoid(lg.to(DEV)).cpu()
        tp, fp, fn, tn = O.seg_counts(p, m, thr, is_logits=False)
        near = int(((p - thr).abs() < 4e-7).sum())               # pixels whose sigmoid is within an ulp or two of thr
        for k, v in (("tp", tp), ("fp", fp), ("fn", fn), ("tn", tn)):
            assert abs(got[k] - 2 * v) <= 2 * near, (thr, k, got[k], 2 * v, near)
        assert got["tp"] + got["fp"] + got["fn"] + got["tn"] == 2 * lg.numel()
        c2 = SegCounts(thr)
        c2.update(p.to(DEV), m.to(DEV), is_logits=False)
        assert (c2.compute()["tp"], c2.compute()["fp"], c2.compute()["fn"], c2.compute()["tn"]) == (tp, fp, fn, tn)


def test_pr_curve_histogram_exact_and_best_threshold():
    from hyperpri_amd.trainer import PRCurve, best_dice_threshold
    n = 2 * 608 * 968
    p = _u(51, (n,)) ** 2
    p[:8] = torch.tensor([0.0, 1.0, 0.5, 0.25, 1.0 / 499, 498.0 / 499, 0.1, 0.998])
    t = (_u(52, (n,)) < p).float()
    pr = PRCurve(500)
    half = n // 2
    pr.update(p[:half].to(DEV), t[:half].to(DEV))
    pr.update(p[half:].to(DEV), t[half:].to(DEV).to(torch.int64))
    tp, fp, fn, tn = pr.confusion()
    oprec, orec, oth, otp, ofp, ofn = O.pr_curve_binned(p, t, 500)
    assert torch.equal(tp, otp) and torch.equal(fp, ofp) and torch.equal(fn, ofn)
    prec, rec, th = pr.compute()
    assert torch.equal(prec, oprec) and torch.equal(rec, orec) and torch.equal(th, oth)
    assert best_dice_threshold(prec, rec, th) == O.best_dice_threshold(oprec, orec, oth)
    # logits in, sigmoid inside the kernel: same curve up to pixels that sit within an ulp of a threshold
    lg = torch.logit(p.clamp(1e-6, 1 - 1e-6))
    pr2 = PRCurve(500)
    pr2.update(lg.to(DEV), t.to(DEV), is_logits=True)
    tp2, fp2, _, _ = pr2.confusion()
    o2 = O.pr_curve_binned(torch.sigmoid(lg), t, 500)
    assert int((tp2 - o2[3]).abs().max()) <= 4 and int((fp2 - o2[4]).abs().max()) <= 4


def _rand_params(seed, sizes):
    return [(_u(seed + i, (s,)) - 0.5) * 0.2 for i, s in enumerate(sizes)]


SIZES = [1, 3, 64, 257, 4096, 4097, 70000, 9 * 64 * 64] + [5 + i for i in range(50)]      # 58 tensors: two launches


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_fused_adam_matches_torch_adam(wd):
    from hyperpri_amd.trainer import FusedAdam
    p0 = _rand_params(100, SIZES)
    grads = [[(_u(1000 * s + i, (n,)) - 0.5) * (10.0 ** -(i % 5)) for i, n in enumerate(SIZES)] for s in range(1, 4)]
    grads[1][3] = None                                           # a tensor without a gradient in step 2 is skipped
    ref = O.optimizer_steps("adam", p0, grads, lr=1e-3, weight_decay=wd)
    ps = [torch.nn.Parameter(p.clone().to(DEV)) for p in p0]
    opt = FusedAdam(ps, lr=1e-3, weight_decay=wd)
    for gs in grads:
        for p, g in zip(ps, gs):
            p.grad = None if g is None else g.to(DEV)
        opt.step()
    for i, (p, r) in enumerate(zip(ps, ref)):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.numpy(), rtol=2e-6, atol=5e-8, err_msg=str(i))
    assert opt.state[ps[3]]["step"] == 2 and opt.state[ps[0]]["step"] == 3


@pytest.mark.parametrize("momentum,wd", [(0.0, 0.0), (0.9, 1e-4)])
def test_fused_sgd_matches_torch_sgd(momentum, wd):
    from hyperpri_amd.trainer import FusedSGD
    p0 = _rand_params(200, SIZES)
    grads = [[(_u(2000 * s + i, (n,)) - 0.5) for i, n in enumerate(SIZES)] for s in range(1, 4)]
    ref = O.optimizer_steps("sgd", p0, grads, lr=1e-2, momentum=momentum, weight_decay=wd)
    ps = [torch.nn.Parameter(p.clone().to(DEV)) for p in p0]
    opt = FusedSGD(ps, lr=1e-2, momentum=momentum, weight_decay=wd)
    for gs in grads:
        for p, g in zip(ps, gs):
            p.grad = g.to(DEV)
        opt.step()
    for i, (p, r) in enumerate(zip(ps, ref)):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.numpy(), rtol=2e-6, atol=5e-8, err_msg=str(i))


def test_adam_grad_scale_is_applied_on_device():
    from hyperpri_amd.trainer import FusedAdam
    p0 = _rand_params(300, [1000])
    g = (_u(301, (1000,)) - 0.5)
    ref = O.optimizer_steps("adam", p0, [[g * 0.25]], lr=1e-3)
    p = torch.nn.Parameter(p0[0].clone().to(DEV)); p.grad = g.to(DEV)
    FusedAdam([p], lr=1e-3).step(grad_scale=torch.tensor(0.25, device=DEV))
    np.testing.assert_allclose(p.detach().cpu().numpy(), ref[0].numpy(), rtol=2e-6, atol=5e-8)


@pytest.mark.parametrize("name", ["step_unet3_tiny_adam", "step_cubenet64_tiny_sgd"])
def test_training_steps_vs_reference_fixture(name):
    """RootLightningModel.training_step semantics end to end on the HIP path: network, loss, metrics, optimizer."""
    import hyperpri_amd as H
    from hyperpri_amd.trainer import SegmentationModel
    z = np.load(os.path.join(G, name + ".npz"))
    m = (_u(4321, (2, 1, 36, 50)) > 0.9).float().to(DEV)
    if "unet3" in name:
        net, x = H.UNet(3, 1, bilinear=False), _u(1234, (2, 3, 36, 50))
        model = SegmentationModel(net, optimizer="Adam", lr=1e-3, weight_decay=0.0)
    else:
        net, x = H.CubeNET(6, 1, first_depth=64, bilinear=False), _u(1235, (2, 1, 6, 36, 50))
        model = SegmentationModel(net, optimizer="SGD", lr=1e-2, momentum=0.9, weight_decay=1e-4)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    model = model.to(DEV).train()
    opt = model.configure_optimizers()
    batch = {"image": x.to(DEV), "mask": m}
    for s in range(3):
        opt.zero_grad()
        loss = model.training_step(batch, s)
        loss.backward()
        opt.step()
        assert abs(float(loss.detach()) - z["loss"][s]) < 2e-5, (s, float(loss.detach()), z["loss"][s])
        met = model.epoch_metrics("tr")
        assert abs(met["tr_loss"] - z["loss"][s]) < 2e-5
        acc, dice, iou = z["metrics"][s]
        if s == 0:       # identical weights: Dice/IoU to 4 dp; later steps compare two training runs on a 3600-pixel
            assert round(met["tr_dice"], 4) == round(dice, 4) and round(met["tr_pos_iou"], 4) == round(iou, 4)   # image,
        assert abs(met["tr_dice"] - dice) < 1e-3 and abs(met["tr_pos_iou"] - iou) < 1e-3     # where one pixel is 3e-4
        assert abs(met["tr_acc"] - acc) < 1e-3
    names = [str(s) for s in z["param_names"]]
    params = OrderedDict(net.named_parameters())
    assert names == list(params.keys())
    for i, k in enumerate(names):
        v = params[k].detach().double().flatten().cpu()
        # Adam normalises by |g|: parameters whose true gradient is 0 (conv bias in front of train-mode BN) move by
        # rounding noise on both sides -- their norms get a looser absolute bound
        loose = 4e-3 if (k.endswith("bias") and v.numel() <= 1024) else 0.0
        assert abs(float(v.norm()) - z["param_l2"][i]) <= 2e-4 * z["param_l2"][i] + 2e-5 + loose, (k, float(v.norm()), z["param_l2"][i])
    net.eval()
    with torch.no_grad():
        le = net(batch["image"]).cpu().numpy()
    assert np.abs(le - z["logits_eval"]).max() < 5e-3


def test_validation_and_predict_steps():
    import hyperpri_amd as H
    from hyperpri_amd.trainer import SegmentationModel
    net = H.UNet(3, 1, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    sd = O.synth_state_dict(shapes)
    net.load_state_dict(sd)
    model = SegmentationModel(net).to(DEV).eval()
    x = _u(1234, (2, 3, 36, 50)); m = (_u(4321, (2, 1, 36, 50)) > 0.9).float()
    batch = {"image": x.to(DEV), "mask": m.to(DEV)}
    with torch.no_grad():
        model.validation_step(batch, 0)
        pred = model.predict_step(batch, 0)
    ref = O.unet_forward(OrderedDict((k, v.clone()) for k, v in sd.items()), x, train=False)
    assert pred.device.type == "cpu" and np.abs(pred.numpy() - ref.numpy()).max() < 1e-3
    assert len(model.predict_labels) == 1 and torch.equal(model.predict_labels[0], m)
    met = model.epoch_metrics("val")
    acc, dice, iou = O.seg_metrics(ref, m)
    assert round(met["val_dice"], 4) == round(dice, 4) and round(met["val_pos_iou"], 4) == round(iou, 4)
    assert abs(met["val_loss"] - float(O.bce_with_logits(ref, m))) < 1e-5


def test_average_precision_matches_sklearn():
    from hyperpri_amd.trainer import average_precision
    n = 200000
    p = (_u(61, (n,)) ** 2)
    p[:1000] = torch.round(p[:1000] * 20) / 20          # plenty of ties
    t = (_u(62, (n,)) < p).float()
    got = average_precision(p.to(DEV), t.to(DEV))
    want = O.average_precision(p, t)
    assert abs(got - want) < 1e-9, (got, want)
    assert average_precision(p.to(DEV), torch.zeros(n, device=DEV)) != average_precision(p.to(DEV), torch.zeros(n, device=DEV))  # nan


def _tiny_head_net():
    import hyperpri_amd as H
    torch.manual_seed(3)
    return H.UNet(3, 1, bilinear=False).to(DEV).train()


def test_forward_loss_checks_what_bcewithlogits_checks():
    """the fused head only compared element counts.  A (N,H,W) mask against (N,1,H,W) logits must raise
    ValueError as nn.BCEWithLogitsLoss does; an in-place write to the logits or the target between forward and backward must raise
    as tensors saved for backward would; a loss that is never back-propagated must not keep the tape alive through a reference
    cycle (memory returns without the cyclic collector)."""
    import gc
    import hyperpri_amd as H
    net = _tiny_head_net()
    x = _u(41, (2, 3, 16, 24)).to(DEV)
    mask = (_u(42, (2, 1, 16, 24)) > 0.5).float().to(DEV)
    with pytest.raises(ValueError, match="must be the same as input size"):
        H.forward_loss(net, x, mask[:, 0])
    # target written in place after the forward
    pred, loss = H.forward_loss(net, x, mask)
    mask.mul_(0.5)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()
    mask = (_u(42, (2, 1, 16, 24)) > 0.5).float().to(DEV)
    # logits written in place after the forward
    pred, loss = H.forward_loss(net, x, mask)
    pred.detach().add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()
    # untouched: same gradients as the two-call form
    for p in net.parameters():
        p.grad = None
    pred, loss = H.forward_loss(net, x, mask)
    loss.backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    H.BCEWithLogitsLoss()(net(x), mask).backward()
    for a, b in zip(g1, [p.grad for p in net.parameters()]):       # (equal up to the summation order inside the head's own gradient)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-8)
    # no cycle: a dropped (pred, loss) pair frees its activations without gc
    del pred, loss, g1
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        for _ in range(3):
            pred, loss = H.forward_loss(net, x, mask)
            del pred, loss
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() <= base + 4096, (torch.cuda.memory_allocated(), base)
    finally:
        gc.enable()
