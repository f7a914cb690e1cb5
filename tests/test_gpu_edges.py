"""Edge cases of the module API on the HIP path vs the CPU oracle: smallest legal images, batch 1 and batch 5, ragged
(odd, prime) geometry at every level, inputs that need a gradient, non-contiguous / channels-last callers, gradient
accumulation over two backward passes, a second forward before backward, and the error behaviour the reference's
callers rely on.  Needs a real MI355X: ``-m gpu``."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hyperpri_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _u(seed, shape):
    return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())


def _pair(kind):
    import hyperpri_amd as H
    if kind == "unet":
        net, shapes, fwd, kw = H.UNet(3, 1, bilinear=False), O.unet_shapes(3, 1), O.unet_forward, {}
    elif kind == "cube":
        net, shapes, fwd, kw = H.CubeNET(5, 1, first_depth=64, bilinear=False), O.cubenet_shapes(5, 1, 64), O.cubenet_forward, dict(first_depth=64)
    else:
        net, shapes, fwd, kw = H.SpectralUNET(7, 1, 12), O.spectral_shapes(7, 1, 12), O.spectral_forward, {}
    sd = O.synth_state_dict(shapes)
    net.load_state_dict(sd)
    return net.to(DEV).train(), sd, fwd, kw


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("kind,shape", [
    ("unet", (1, 3, 16, 16)),        # 1x1 bottleneck with batch 1: train-mode BatchNorm raises, as in torch
    ("unet", (2, 3, 16, 16)),        # smallest size that trains: two values per channel at the bottleneck
    ("unet", (2, 3, 17, 31)),        # odd, prime: floor at every pool, right/bottom pad at every up
    ("unet", (5, 3, 20, 44)),        # batch not a power of two
    ("cube", (2, 1, 5, 16, 16)),
    ("cube", (3, 1, 5, 23, 37)),
    ("spec", (1, 7, 1, 1)),          # a single pixel per image: BatchNorm1d over one row must raise like torch
    ("spec", (2, 7, 3, 5)),
])
def test_small_and_ragged_geometry_vs_oracle(kind, shape):
    net, sd, fwd, kw = _pair(kind)
    x = _u(900, shape)
    hw = shape[-2:]
    mask = (_u(901, (shape[0], 1) + tuple(hw)) > 0.7).float()
    work = OrderedDict((k, v.clone()) for k, v in sd.items())
    if (kind == "spec" and hw == (1, 1)) or (kind == "unet" and shape == (1, 3, 16, 16)):
        with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
            O.train_step(fwd, work, x, mask, **kw)
        with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
            net(x.to(DEV))
        net.eval()                               # eval mode has no such limit; the layers in front of the failing
        with torch.no_grad():                    # BatchNorm have updated their running stats on both sides
            le = net(x.to(DEV)).cpu()
            ref = fwd(OrderedDict((k, v.detach().clone()) for k, v in work.items()), x, train=False, **kw)
        assert np.abs(le.numpy() - ref.numpy()).max() < 1e-3
        return
    ref_logits, ref_loss, ref_grads = O.train_step(fwd, work, x, mask, **kw)
    logits = net(x.to(DEV))
    loss = torch.nn.BCEWithLogitsLoss()(logits, mask.to(DEV))
    loss.backward()
    assert logits.shape == ref_logits.shape
    # a 1x1 bottleneck batch-normalises 1-2 values per channel: ill-conditioned on both sides -> relative L2
    assert _rel(logits.detach().cpu(), ref_logits) < 3e-2
    if min(hw) >= 20:
        assert np.abs(logits.detach().cpu().numpy() - ref_logits.numpy()).max() < 1e-3
        assert abs(float(loss.detach()) - ref_loss) < 1e-5
        for (k, p) in net.named_parameters():
            r = ref_grads[k]
            if float(r.norm()) > 1e-6:
                assert _rel(p.grad.cpu(), r) < 8e-2, k   # BN over <= 10 bottleneck values amplifies rounding on both sides (oracle fp32 vs fp64: 3.6 %)


def test_input_gradient_and_noncontiguous_callers():
    net, sd, fwd, kw = _pair("unet")
    x = _u(910, (2, 3, 24, 40))
    mask = (_u(911, (2, 1, 24, 40)) > 0.8).float()
    xr = x.clone().requires_grad_(True)
    work = OrderedDict((k, v.clone()) for k, v in sd.items())
    for k in work:
        if O.is_param(k):
            work[k] = work[k].requires_grad_(True)
    O.bce_with_logits(fwd(work, xr, train=True), mask).backward()
    # (a) leaf input that needs a gradient
    xg = x.to(DEV).requires_grad_(True)
    torch.nn.BCEWithLogitsLoss()(net(xg), mask.to(DEV)).backward()
    assert xg.grad is not None and xg.grad.shape == x.shape and _rel(xg.grad.cpu(), xr.grad) < 5e-3
    # (b) a transposed (non-contiguous) view and (c) a channels-last tensor give the same logits as the plain tensor
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.reset_running_stats()
    with torch.no_grad():
        base = net(x.to(DEV))
        nc = net(x.to(DEV).transpose(2, 3).contiguous().transpose(2, 3))
        cl = net(x.to(DEV).contiguous(memory_format=torch.channels_last))
    assert torch.equal(base, nc) and torch.equal(base, cl)


def test_gradient_accumulation_and_two_forwards_before_backward():
    net, sd, fwd, kw = _pair("cube")
    xa, xb = _u(920, (2, 1, 5, 24, 40)).to(DEV), _u(921, (2, 1, 5, 24, 40)).to(DEV)
    ma, mb = (_u(922, (2, 1, 24, 40)) > 0.8).float().to(DEV), (_u(923, (2, 1, 24, 40)) > 0.8).float().to(DEV)
    crit = torch.nn.BCEWithLogitsLoss()

    def reset():
        for p in net.parameters():
            p.grad = None
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.reset_running_stats()
    reset()
    crit(net(xa), ma).backward()
    ga = [p.grad.clone() for p in net.parameters()]
    reset()
    crit(net(xb), mb).backward()
    gb = [p.grad.clone() for p in net.parameters()]
    # accumulate over two backward passes without zeroing (p.grad += ...)
    reset()
    crit(net(xa), ma).backward()
    crit(net(xb), mb).backward()
    for p, a, b in zip(net.parameters(), ga, gb):
        assert float((p.grad - (a + b)).abs().max()) <= 1e-6 * float((a + b).abs().max()) + 1e-12
    # two forwards, then both backwards (each graph keeps its own saved activations)
    reset()
    la, lb = crit(net(xa), ma), crit(net(xb), mb)
    lb.backward()
    la.backward()
    for p, a, b in zip(net.parameters(), ga, gb):
        assert float((p.grad - (a + b)).abs().max()) <= 1e-6 * float((a + b).abs().max()) + 1e-12
    # backward twice through the same graph is an error, as for any torch graph without retain_graph
    reset()
    l = crit(net(xa), ma)
    l.backward()
    with pytest.raises(RuntimeError):
        l.backward()


def test_errors_the_callers_would_see():
    import hyperpri_amd as H
    net = H.CubeNET(5, 1, first_depth=64, bilinear=False).to(DEV)
    with pytest.raises(ValueError):
        net(torch.zeros(1, 1, 6, 16, 16, device=DEV))          # wrong number of bands
    with pytest.raises(ValueError):
        net(torch.zeros(1, 5, 16, 16, device=DEV))             # not unsqueezed (dataset.py:269-270)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 5, 16, 16, device=DEV, dtype=torch.float16))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 5, 16, 16, device=DEV, dtype=torch.float64))
    unet = H.UNet(3, 1, bilinear=False).to(DEV)
    with pytest.raises((RuntimeError, ValueError)):
        unet(torch.zeros(1, 4, 16, 16, device=DEV))            # channel mismatch
    with pytest.raises((RuntimeError, ValueError)):
        unet(torch.zeros(1, 3, 8, 8, device=DEV))              # 8x8: the fourth pool has nothing to pool (torch raises too)


def test_frozen_parameters_get_no_gradient_and_the_rest_is_unchanged():
    """feature_extraction / set_parameter_requires_grad (models.py:279-292) freezes parameters: frozen ones must end
    with grad None, the others with exactly the gradient of the unfrozen run."""
    net, sd, fwd, kw = _pair("unet")
    x = _u(930, (2, 3, 24, 40)).to(DEV)
    mask = (_u(931, (2, 1, 24, 40)) > 0.8).float().to(DEV)
    crit = torch.nn.BCEWithLogitsLoss()

    def reset():
        for p in net.parameters():
            p.grad = None
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.reset_running_stats()
    reset()
    crit(net(x), mask).backward()
    full = {k: p.grad.clone() for k, p in net.named_parameters()}
    reset()
    frozen = [k for k, _ in net.named_parameters() if k.startswith(("inc.", "down1.", "down2."))]
    for k, p in net.named_parameters():
        p.requires_grad_(k not in frozen)
    crit(net(x), mask).backward()
    for k, p in net.named_parameters():
        if k in frozen:
            assert p.grad is None, k
        else:
            assert torch.equal(p.grad, full[k]), k
    # everything frozen + an input that needs no gradient: forward still works and builds no graph
    for p in net.parameters():
        p.requires_grad_(False)
    out = net(x)
    assert not out.requires_grad
