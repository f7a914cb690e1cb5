"""HIP modules (through the C ABI) against the golden fixtures captured from the reference modules and
against the CPU oracle.  Needs a real MI355X: run with ``-m gpu``."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hyperpri_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _load(name):
    return np.load(os.path.join(G, name + ".npz"))


def _shapes(mod):
    return OrderedDict((k, tuple(v.shape)) for k, v in mod.state_dict().items())


def _close(a, b, rtol, atol, msg=""):
    a = a.detach().cpu().contiguous().numpy() if torch.is_tensor(a) else a
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def _block(name, mod, seed0, nin):
    import hyperpri_amd  # noqa: F401
    z = _load(name)
    mod.load_state_dict(O.synth_state_dict(_shapes(mod), seed0=seed0, bn_random=True))
    mod = mod.to(DEV).train()
    xs = [torch.from_numpy(z[f"in{i}"]).to(DEV).requires_grad_(True) for i in range(nin)]
    out = mod(*xs)
    assert out.shape == z["out_train"].shape
    _close(out, z["out_train"], 1e-4, 2e-5, "train forward")
    (out * torch.from_numpy(z["dout"]).to(DEV)).sum().backward()
    for i, x in enumerate(xs):
        _close(x.grad, z[f"din{i}"], 1e-3, 2e-5, f"input grad {i}")
    for k, p in mod.named_parameters():
        ref = z["grad/" + k]
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            # a conv bias in front of a training-mode BN has a mathematically ZERO gradient: both sides
            # hold only fp32 rounding noise of a sum over all pixels, so compare against zero
            assert float(p.grad.abs().max()) < 1e-3 and float(np.abs(ref).max()) < 1e-3, k
            continue
        _close(p.grad, ref, 1e-3, 3e-5 * max(1.0, float(np.abs(ref).max())), "grad " + k)
    for k, b in mod.named_buffers():
        _close(b.float() if b.dtype != torch.float32 else b, z["buf/" + k].astype(np.float32), 1e-4, 1e-5, "buffer " + k)
    mod.eval()
    with torch.no_grad():
        oe = mod(*[x.detach() for x in xs])
    _close(oe, z["out_eval"], 1e-4, 2e-5, "eval forward")


def test_doubleconv():
    from hyperpri_amd import DoubleConv
    _block("block_doubleconv", DoubleConv(5, 7), 2000, 1)


def test_doubleconv_mid():
    from hyperpri_amd import DoubleConv
    _block("block_doubleconv_mid", DoubleConv(6, 4, 9), 2100, 1)


def test_down():
    from hyperpri_amd import Down
    _block("block_down", Down(4, 6), 2200, 1)


def test_up():
    from hyperpri_amd import Up
    _block("block_up", Up(8, 4, bilinear=False), 2300, 2)


def test_up_big():
    from hyperpri_amd import Up
    _block("block_up_big", Up(64, 32, bilinear=False), 2400, 2)


@pytest.mark.parametrize("name,bil,att,seed", [("block_up_bilinear", True, False, 2600), ("block_up_attn", False, True, 2700),
                                               ("block_up_bilinear_attn", True, True, 2800)])
def test_up_variants(name, bil, att, seed):
    """Up(bilinear=True) and Up(use_attention=True): SURVEY.md section 8 a3' (not configured by any experiment)."""
    from hyperpri_amd import Up
    _block(name, Up(16, 8, bilinear=bil, use_attention=att), seed, 2)


@pytest.mark.parametrize("name,bil,seed", [("block_up_crop", False, 3100), ("block_up_crop_mixed", False, 3200),
                                           ("block_up_bilinear_crop", True, 3300)])
def test_up_crop(name, bil, seed):
    """Skip smaller than the upsampled tensor (F.pad with negative widths crops, model_parts.py:73-80): both axes cropped,
    one cropped and one padded, uneven crops after bilinear upsampling."""
    from hyperpri_amd import Up
    _block(name, Up(16, 8, bilinear=bil), seed, 2)


def test_outconv():
    from hyperpri_amd import OutConv
    _block("block_outconv", OutConv(6, 2), 2500, 1)


@pytest.mark.parametrize("N,Cin,H,W,Cout,ks", [(2, 5, 9, 11, 7, 3), (1, 64, 20, 70, 128, 3), (2, 40, 13, 37, 64, 3),
                                               (1, 24, 5, 33, 200, 1), (2, 256, 9, 40, 64, 3)])
def test_conv_fwd_bwd_vs_torch(N, Cin, H, W, Cout, ks):
    """Bare conv (no BN) forward, data-gradient and weight/bias gradients vs torch CPU conv2d."""
    from hyperpri_amd import engine as E
    from hyperpri_amd.autograd import run
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(N * 1000 + Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    xc, wc, bc = (t.clone().requires_grad_(True) for t in (x, w, b))
    yc = F.conv2d(xc, wc, bc, padding=ks // 2)
    (yc * r).sum().backward()
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = run(lambda tape, a, need: E.conv_bn_relu(tape, a[0], wd, bd, None, True, ks, need_dx=need[0]), [xd], [wd, bd])
    _close(yd, yc.detach().numpy(), 1e-4, 1e-4, "conv forward")
    (yd * r.to(DEV)).sum().backward()
    _close(xd.grad, xc.grad.numpy(), 1e-4, 2e-4, "conv dgrad")
    _close(wd.grad, wc.grad.numpy(), 1e-4, 1e-3, "conv wgrad")
    _close(bd.grad, bc.grad.numpy(), 1e-4, 1e-3, "conv bias grad")


def test_cpu_tensor_raises():
    from hyperpri_amd import DoubleConv
    m = DoubleConv(4, 4)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 8, 8))


@pytest.mark.parametrize("N,Cin,H,W,Cout,ks", [(2, 5, 9, 11, 7, 3), (1, 64, 20, 70, 128, 3), (2, 40, 13, 37, 64, 3),
                                               (1, 24, 5, 33, 200, 1), (2, 256, 9, 40, 64, 3), (1, 512, 38, 60, 512, 3)])
def test_conv_bf16_mode_vs_torch_on_rounded_operands(N, Cin, H, W, Cout, ks):
    """precision="bf16": the kernel rounds activations and weights to bf16 (RNE) and accumulates in fp32, so it must
    agree with torch's fp32 conv evaluated on the bf16-rounded operands to fp32-summation accuracy (forward and data
    and weight gradients)."""
    from hyperpri_amd import engine as E
    from hyperpri_amd.autograd import run
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(N * 1000 + Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    xc, wq = rb(x).requires_grad_(True), rb(w).requires_grad_(True)
    yc = F.conv2d(xc, wq, b, padding=ks // 2)
    yc.backward(rb(r))                       # dgrad / wgrad operands (dy, w, x) are rounded too
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = run(lambda tape, a, need: E.conv_bn_relu(tape, a[0], wd, bd, None, True, ks, need_dx=need[0], precision="bf16"),
             [xd], [wd, bd])
    _close(yd, yc.detach().numpy(), 1e-4, 2e-4, "bf16 conv forward")
    yd.backward(r.to(DEV))
    _close(xd.grad, xc.grad.numpy(), 1e-4, 3e-4, "bf16 conv dgrad")
    _close(wd.grad, wq.grad.numpy(), 1e-4, 2e-3, "bf16 conv wgrad")


def test_up_block_bf16_mode_vs_fixture():
    """Up (ConvTranspose2d scatter, concat, DoubleConv) with every contraction in bf16 mode: segmentation-level
    tolerance against the fp32 reference fixture (ConvTranspose2d forward / data / weight gradients on the bf16 kernels)."""
    import hyperpri_amd as H
    z = _load("block_up_big")
    mod = H.Up(64, 32, bilinear=False)
    mod.load_state_dict(O.synth_state_dict(_shapes(mod), seed0=2400, bn_random=True))
    mod = H.set_precision(mod.to(DEV), "bf16").train()
    xs = [torch.from_numpy(z[f"in{i}"]).to(DEV).requires_grad_(True) for i in range(2)]
    out = mod(*xs)
    ref = z["out_train"]
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 0.05 * max(1.0, float(np.abs(ref).max()))
    (out * torch.from_numpy(z["dout"]).to(DEV)).sum().backward()
    for i, x in enumerate(xs):
        r = z[f"din{i}"]
        assert np.linalg.norm(x.grad.cpu().numpy() - r) <= 0.12 * np.linalg.norm(r)   # two BN layers amplify bf16 rounding
    for k in ("up.weight", "conv.double_conv.0.weight", "conv.double_conv.3.weight"):
        r = z["grad/" + k]
        g = dict(mod.named_parameters())[k].grad.cpu().numpy()
        assert np.linalg.norm(g - r) <= 0.12 * np.linalg.norm(r), k


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_transposed_conv_concat_vs_torch(prec):
    """ConvTranspose2d(k2,s2) + right/bottom zero pad + concat (engine.up_concat) against torch on CPU, forward and all
    gradients.  In bf16 mode the reference uses bf16-rounded operands, so the tolerance stays at fp32-summation level."""
    from hyperpri_amd import engine as E
    from hyperpri_amd.autograd import run
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(99)
    N, Cin, H, W, Cs = 2, 64, 7, 19, 32
    x1 = torch.randn(N, Cin, H, W, generator=g)
    sk = torch.randn(N, Cs, 2 * H + 1, 2 * W + 1, generator=g)
    w = torch.randn(Cin, Cin // 2, 2, 2, generator=g) / (Cin * 4) ** 0.5
    b = torch.randn(Cin // 2, generator=g)
    r = torch.randn(N, Cs + Cin // 2, 2 * H + 1, 2 * W + 1, generator=g)
    rb = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if prec == "bf16" else (lambda t: t.clone())
    x1c, wc, bc, skc = rb(x1).requires_grad_(True), rb(w).requires_grad_(True), b.clone().requires_grad_(True), sk.clone().requires_grad_(True)
    up = F.pad(F.conv_transpose2d(x1c, wc, bc, stride=2), [0, 1, 0, 1])
    yc = torch.cat([skc, up], 1)
    rr = r.clone()
    rr[:, Cs:] = rb(r[:, Cs:])            # the gradient reaching the transposed conv is rounded as an operand
    yc.backward(rr)
    x1d, skd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x1, sk, w, b))
    yd = run(lambda tape, a, need: E.up_concat(tape, a[0], a[1], wd, bd, need_dx1=need[0], precision=prec), [x1d, skd], [wd, bd])
    _close(yd, yc.detach().numpy(), 1e-4, 2e-4, "convT+cat forward")
    yd.backward(r.to(DEV))
    _close(x1d.grad, x1c.grad.numpy(), 1e-4, 3e-4, "convT dgrad")
    _close(skd.grad, r[:, :Cs].numpy(), 0, 0, "skip gradient is the channel slice")
    _close(wd.grad, wc.grad.numpy(), 1e-4, 2e-3, "convT wgrad")
    # the bias gradient is a plain fp32 column sum of the (unrounded) incoming gradient over the un-padded region
    _close(bd.grad, r[:, Cs:, :2 * H, :2 * W].sum(dim=(0, 2, 3)).numpy(), 1e-4, 1e-3, "convT bias grad")


@pytest.mark.parametrize("N,Cin,H,W,Cout,ks", [(2, 5, 9, 11, 7, 3), (1, 64, 20, 70, 128, 3), (2, 40, 13, 37, 64, 3),
                                               (1, 24, 5, 33, 200, 1), (1, 512, 38, 60, 512, 3)])
def test_conv_bf16x3_mode_vs_exact(N, Cin, H, W, Cout, ks):
    """precision="bf16x3": operands carried as bf16 hi + bf16 lo (16 mantissa bits), hi*hi + hi*lo + lo*hi on the bf16
    pipe.  Against the EXACT fp32 torch conv: relative error of a product ~2^-16, i.e. ~1e-5 of the output scale."""
    from hyperpri_amd import engine as E
    from hyperpri_amd.autograd import run
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(N * 1000 + Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yc = F.conv2d(xc.double(), wc.double(), b.double(), padding=ks // 2)
    yc.backward(r.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = run(lambda tape, a, need: E.conv_bn_relu(tape, a[0], wd, bd, None, True, ks, need_dx=need[0], precision="bf16x3"),
             [xd], [wd, bd])
    sc = float(yc.detach().abs().max())
    assert float((yd.detach().cpu().double() - yc.detach()).abs().max()) < 1e-4 * sc
    yd.backward(r.to(DEV))
    assert float((xd.grad.cpu().double() - xc.grad).abs().max()) < 1e-4 * float(xc.grad.abs().max())
    assert float((wd.grad.cpu().double() - wc.grad).abs().max()) < 1e-4 * float(wc.grad.abs().max())


@pytest.mark.parametrize("N,Cin,H,W,Cout,ks", [(2, 5, 9, 11, 7, 3), (1, 64, 20, 70, 128, 3), (2, 40, 13, 37, 64, 3),
                                               (1, 24, 5, 33, 200, 1), (1, 512, 38, 60, 512, 3), (2, 238, 24, 40, 64, 3)])
def test_conv_bf16x6_mode_is_as_exact_as_fp32(N, Cin, H, W, Cout, ks):
    """precision="bf16x6": three bf16 planes = the fp32 operand exactly, six MFMAs per product.  Against the fp64
    convolution it must be (almost) as close as the exact fp32 MFMA path is: both are measured and compared."""
    from hyperpri_amd import engine as E
    from hyperpri_amd.autograd import run
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(N * 1000 + Cin + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(N, Cout, H, W, generator=g)
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yc = F.conv2d(xc.double(), wc.double(), b.double(), padding=ks // 2)
    yc.backward(r.double())
    err = {}
    for prec in ("fp32", "bf16x6"):
        xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        yd = run(lambda tape, a, need: E.conv_bn_relu(tape, a[0], wd, bd, None, True, ks, need_dx=need[0], precision=prec),
                 [xd], [wd, bd])
        yd.backward(r.to(DEV))
        err[prec] = (float((yd.detach().cpu().double() - yc.detach()).abs().max()) / float(yc.detach().abs().max()),
                     float((xd.grad.cpu().double() - xc.grad).abs().max()) / float(xc.grad.abs().max()),
                     float((wd.grad.cpu().double() - wc.grad).abs().max()) / float(wc.grad.abs().max()))
    for e6, e32 in zip(err["bf16x6"], err["fp32"]):
        assert e6 < 2e-6, err                      # ~fp32 rounding level (bf16x3 sits at 1e-5)
        assert e6 <= 4.0 * e32 + 2e-7, err         # and never far from what the fp32 MFMA path itself achieves
