"""GPU part of tests/test_dispatcher.py: a module call really goes through ``torch.ops.hyperpri.<name>`` (seen by a
TorchDispatchMode), and the operator path gives bit-identical logits, loss and gradients to the plain autograd.Function path
(HPRI_DISPATCHER=0) on the network-level and block-level modules.  Needs a real MI355X: ``-m gpu``."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import hyperpri_amd as H
from hyperpri_amd import autograd as A
from oracle import hyperpri_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _u(seed, shape):
    return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())


class _Seen(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.names = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        self.names.append(str(func))
        return func(*args, **(kwargs or {}))


def _step(net, x, mask, dispatcher):
    old = A.USE_DISPATCHER
    A.USE_DISPATCHER = dispatcher
    try:
        for p in net.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        out = net(xi) if not isinstance(x, tuple) else net(*x)
        loss = H.BCEWithLogitsLoss()(out, mask) if mask is not None else out.square().mean()
        loss.backward()
        torch.cuda.synchronize()
        return out.detach().clone(), float(loss), [p.grad.clone() for p in net.parameters()], xi.grad.clone()
    finally:
        A.USE_DISPATCHER = old


@pytest.mark.parametrize("kind", ["unet", "cubenet", "spectral_unet", "double_conv", "down"])
def test_operator_path_is_bit_identical_to_the_function_path(kind):
    torch.manual_seed(5)
    if kind == "unet":
        net, x, mask = H.UNet(3, 1, bilinear=False), _u(7, (2, 3, 16, 24)), (_u(8, (2, 1, 16, 24)) > 0.5).float()
    elif kind == "cubenet":
        net, x, mask = H.CubeNET(6, 1, first_depth=64, bilinear=False), _u(7, (2, 1, 6, 16, 24)), (_u(8, (2, 1, 16, 24)) > 0.5).float()
    elif kind == "spectral_unet":
        net, x, mask = H.SpectralUNET(10, 1, 4), _u(7, (3, 10, 4, 5)), (_u(8, (3, 1, 4, 5)) > 0.5).float()
    elif kind == "double_conv":
        net, x, mask = H.DoubleConv(5, 7), _u(7, (2, 5, 9, 11)), None
    else:
        net, x, mask = H.Down(8, 12), _u(7, (2, 8, 10, 12)), None
    net = net.to(DEV).train()
    x = x.to(DEV)
    mask = None if mask is None else mask.to(DEV)
    sd = OrderedDict((k, v.clone()) for k, v in net.state_dict().items())
    with _Seen() as seen:
        a = _step(net, x, mask, True)
    assert any(n.startswith(f"hyperpri.{kind}") for n in seen.names), sorted(set(seen.names))[:20]
    net.load_state_dict(sd)                  # BN running statistics back to where the first step found them
    with _Seen() as seen2:
        b = _step(net, x, mask, False)
    assert not any(n.startswith("hyperpri.") for n in seen2.names)
    assert torch.equal(a[0], b[0]) and a[1] == b[1] and torch.equal(a[3], b[3])
    for g1, g2 in zip(a[2], b[2]):
        assert torch.equal(g1, g2)


def test_forward_loss_and_no_grad_through_the_operator():
    net = H.UNet(3, 1, bilinear=False).to(DEV).train()
    x, mask = _u(7, (2, 3, 16, 24)).to(DEV), (_u(8, (2, 1, 16, 24)) > 0.5).float().to(DEV)
    assert A.USE_DISPATCHER
    pred, loss = H.forward_loss(net, x, mask)
    loss.backward()
    g = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    sd_before = net.state_dict()["inc.double_conv.1.num_batches_tracked"].item()
    H.BCEWithLogitsLoss()(net(x), mask).backward()
    for g1, p in zip(g, net.parameters()):          # (the head's own gradient sums in another order in the fused-loss form)
        torch.testing.assert_close(g1, p.grad, rtol=1e-5, atol=1e-8)
    assert net.state_dict()["inc.double_conv.1.num_batches_tracked"].item() == sd_before + 1
    with torch.no_grad():
        y = net.eval()(x)
    assert not y.requires_grad and y.shape == (2, 1, 16, 24)
    with pytest.raises(RuntimeError, match="backward called twice|second time"):
        net.train()
        out = net(x)
        out.sum().backward()
        out.sum().backward()
