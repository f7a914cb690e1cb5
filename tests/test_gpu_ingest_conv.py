"""The predict path of the 16-bit modes (eval mode under inference_mode / no_grad: PLTrainer.py:530-532,626), round 5.
(1) First-layer fused ingest (csrc/conv_ingest.hip, ``hpri_conv3x3_ingest_h16``; reference models.py:169 as called at models.py:215-216):
CubeNET's first 3x3 layer reads the caller's NC(D)HW fp32 cube itself instead of planes written by a layout pass.  Gates: the kernel
against the oracle's arithmetic (fp64 convolution of the rounded operands) and BIT-identical to the pair it replaces (layout pass +
hpri_conv_bf16v3) on ragged shapes, channel counts off every multiple, 1..3 images, 32 / 64 / 96 / 128 outputs; CubeNET's predict forward
with and without it: identical logits; the reference's eval-mode fixture.  (2) Skips and decoder concats as planes only, transposed
convolutions on the plane GEMM (``engine.PREDICT_SKIP_PLANES``).  (3) SpectralUNET's folded Linear -> BatchNorm1d -> ReLU layers on the
plane GEMM with plane concats (``engine.PREDICT_GEMM_PLANES``; models.py:105-115,139-143), against the reference's eval-mode fixtures.
Needs a real MI355X: ``-m gpu``."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from conftest import record_margin
from oracle import hyperpri_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _u(seed, shape):
    return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())


def _bn(cout, seed):
    bn = torch.nn.BatchNorm2d(cout).to(DEV).eval()
    with torch.no_grad():
        bn.weight.copy_(0.5 + _u(seed, (cout,)))
        bn.bias.copy_(_u(seed + 1, (cout,)) - 0.5)
        bn.running_mean.copy_(_u(seed + 2, (cout,)) - 0.5)
        bn.running_var.copy_(0.5 + _u(seed + 3, (cout,)))
    return bn


def _rows(y, cout):
    """the 16-bit rows of a planes-only result as a float64 (P, cout) array (+ the raw bits)"""
    raw = y.pl.buf[: y.P * y.pl.cs].view(y.P, y.pl.cs)[:, y.pl.coff: y.pl.coff + cout]
    return raw


CASES = [  # N, C, H, W, Cout
    (1, 32, 8, 32, 64),        # one tile, one chunk
    (2, 40, 37, 50, 64),       # ragged rows and columns, a half-empty second chunk
    (1, 33, 9, 33, 32),        # one channel / row / column over; 32 outputs (half a block)
    (3, 238, 72, 330, 64),     # the cube's channel count (7 chunks + 14 channels), three images (large enough that the pair does not split K)
    (2, 300, 64, 290, 128),    # C5's channel count, two 64-channel blocks
    (1, 64, 64, 31, 96),       # a single, partial column of tiles; 96 outputs
]


@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("N,C,H,W,cout", CASES, ids=[f"N{c[0]}_C{c[1]}_{c[2]}x{c[3]}_F{c[4]}" for c in CASES])
def test_ingest_kernel_equals_layout_pass_plus_plane_conv_and_the_oracle(N, C, H, W, cout, kind):
    from hyperpri_amd import _lib, engine as E
    x = (_u(100 + C + H, (N, C, H, W)) - 0.3).to(DEV)
    w = ((_u(200 + C, (cout, C, 3, 3)) - 0.5) * (2.0 / np.sqrt(9 * C))).to(DEV)
    b = (_u(300 + C, (cout,)) - 0.5).to(DEV)
    bn = _bn(cout, 400 + C)
    ref = E.BNRef(bn)
    with _lib.using(None if kind == "bf16" else "f16"):
        lazy = E.Act.raw_nchw(x, -1)
        assert lazy is not None
        n0 = E.INGEST_LAUNCHES
        y1 = E._conv_ingest_eval(lazy, w, b, ref, C, cout, True)
        assert E.INGEST_LAUNCHES == n0 + 1
        y0 = E._conv_folded_eval(E.Act.from_tensor(x, -1), w, b, ref, 3, C, cout, True, prec="bf16", inner=cout)
        torch.cuda.synchronize()
    assert not y1.f32_valid
    r1 = _rows(y1, cout)
    assert not y0.f32_valid, "the unfused pair took its split-K form at this size: pick a case where it does not"
    assert torch.equal(r1.view(torch.int16), _rows(y0, cout).view(torch.int16))      # bit for bit the pair it replaces
    # the oracle's arithmetic: operands rounded to the 16-bit type, products and sums exact (fp64), folded eval-mode BatchNorm, ReLU
    dt = torch.bfloat16 if kind == "bf16" else torch.float16
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().cpu().double()
    shift = ((b.cpu().double() - bn.running_mean.cpu().double()) * scale + bn.bias.detach().cpu().double())
    wq = (w.cpu() * scale.float().view(-1, 1, 1, 1)).to(dt).double()
    xq = x.cpu().to(dt).double()
    want = torch.relu(torch.nn.functional.conv2d(xq, wq, shift, padding=1)).permute(0, 2, 3, 1).reshape(-1, cout)
    got = r1.view(dt).double().cpu()
    err = float((got - want).abs().max())
    tol = 2.0 ** (-8 if kind == "bf16" else -11) * max(1.0, float(want.abs().max())) * 1.01 + 1e-3     # one rounding of the result + fp32 summation
    record_margin(f"ingest_conv/{kind}/N{N}_C{C}_{H}x{W}_F{cout}", err, tol)
    assert err <= tol, (err, tol)


def test_ingest_entry_point_refuses_what_it_cannot_do():
    from hyperpri_amd import _lib
    import ctypes
    lib = _lib.load()
    x = torch.zeros(1, 32, 8, 32, device=DEV)
    wp = torch.zeros(9 * 64 * 32, dtype=torch.bfloat16, device=DEV)
    y = torch.zeros(8 * 32 * 64, dtype=torch.bfloat16, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    ok = lambda *a: lib.hpri_conv3x3_ingest_h16(*a)
    assert ok(p(x), p(wp), None, p(y), 64, 0, 1, 32, 8, 32, 64, 64, 1, None) == 0
    assert ok(None, p(wp), None, p(y), 64, 0, 1, 32, 8, 32, 64, 64, 1, None) != 0           # null pointer
    assert ok(p(x), p(wp), None, p(y), 64, 0, 1, 32, 8, 32, 62, 64, 1, None) != 0           # Cout not a multiple of 4
    assert ok(p(x), p(wp), None, p(y), 62, 0, 1, 32, 8, 32, 64, 64, 1, None) != 0           # rows too narrow / misaligned
    assert ok(p(x), p(wp), None, p(y), 64, 0, 1, 32, 8, 32, 64, 96, 1, None) != 0           # Cout_pad not a multiple of 64
    assert ok(p(x), p(wp), None, p(y), 64, 0, 1, 1 << 20, 1 << 10, 1 << 10, 64, 64, 1, None) != 0   # an image beyond 2 GiB
    torch.cuda.synchronize()


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_cubenet_predict_takes_the_fused_ingest_and_changes_nothing(prec):
    """CubeNET's predict forward (eval, inference_mode; PLTrainer.py:530-532) in the 16-bit modes: the first layer reads the caller's
    cube; logits identical to the run with the layout pass, eval-mode fixture within the mode's band, no layout kernel launched."""
    import hyperpri_amd as H
    from hyperpri_amd import engine as E
    z = np.load(os.path.join(G, "net_cubenet64_tiny.npz"))
    net = H.CubeNET(6, 1, first_depth=64, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = net.to(DEV).train()
    x = _u(1235, (2, 1, 6, 36, 50)).to(DEV)
    net(x)                                       # one training forward: the fixture's running statistics
    H.set_precision(net, prec).eval()
    # 6 bands are too few to matter: the lazy input is refused and the layout pass runs as before
    n0 = E.INGEST_LAUNCHES
    with torch.inference_mode():
        le = net(x).cpu().numpy()
    assert E.INGEST_LAUNCHES == n0
    assert float(np.abs(le - z["logits_eval"]).max()) < (0.15 if prec == "bf16" else 0.03)
    # a 40-band cube of the same network family
    net = H.CubeNET(40, 1, first_depth=64, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = H.set_precision(net.to(DEV), prec).train()
    x = _u(1236, (2, 1, 40, 37, 50)).to(DEV)
    with torch.no_grad():
        net(x)
    net.eval()
    E.enable_event_log(True)
    with torch.inference_mode():
        la = net(x)
    torch.cuda.synchronize()
    tags_a = set(E.event_log_summary())
    E.enable_event_log(False)
    assert E.INGEST_LAUNCHES == n0 + 1 and any(t.startswith("conv_ingest_h16") for t in tags_a)
    E.INGEST_FUSED = False
    try:
        with torch.inference_mode():
            lb = net(x)
        assert E.INGEST_LAUNCHES == n0 + 1
    finally:
        E.INGEST_FUSED = True
    assert torch.equal(la, lb)
    # training-mode and grad-mode forwards never see the raw cube
    net.train()
    n1 = E.INGEST_LAUNCHES
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    with torch.no_grad():                        # no tape, but BatchNorm in training mode: batch statistics, nothing to fold
        lt = net(x)
    net.load_state_dict(sd)
    E.INGEST_FUSED = False
    try:
        with torch.no_grad():
            lt0 = net(x)
    finally:
        E.INGEST_FUSED = True
    assert E.INGEST_LAUNCHES == n1 and torch.equal(lt, lt0)
    net.load_state_dict(sd)
    net(x).sum().backward()
    net.eval()
    net(x)                                       # eval, but gradients on: the tape records, so the layout pass runs
    assert E.INGEST_LAUNCHES == n1


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_predict_path_keeps_skips_and_concats_as_planes(prec):
    """Predict path of the 16-bit modes (round 5): a skip tensor is written as 16-bit rows into the decoder concat's plane buffer and
    nowhere else, max-pooling reads those rows, the transposed convolution runs on the plane GEMM and adds its half to the same planes
    (as the training forward does).  Same values as the fp32-skip form up to the summation order of the transposed convolution;
    eval-mode fixture of the reference within the mode's band."""
    import hyperpri_amd as H
    from hyperpri_amd import engine as E
    z = np.load(os.path.join(G, "net_cubenet64_tiny.npz"))
    net = H.CubeNET(6, 1, first_depth=64, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = net.to(DEV).train()
    x = _u(1235, (2, 1, 6, 36, 50)).to(DEV)
    net(x)                                       # one training forward: the fixture's running statistics
    H.set_precision(net, prec).eval()
    with torch.inference_mode():
        le = net(x).cpu().numpy()
    d = float(np.abs(le - z["logits_eval"]).max())
    record_margin(f"predict_planes/{prec}/tiny_fixture", d, 0.15 if prec == "bf16" else 0.03)
    assert d < (0.15 if prec == "bf16" else 0.03)
    # a larger cube: the upper levels are wide enough for the plane kernels' single-pass form
    net = H.CubeNET(40, 1, first_depth=64, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = H.set_precision(net.to(DEV), prec).train()
    x = _u(1237, (2, 1, 40, 128, 192)).to(DEV)
    with torch.no_grad():
        net(x)
    net.eval()
    out, tags, conv = {}, {}, {}
    for flag in (True, False):
        E.PREDICT_SKIP_PLANES = flag
        try:
            c0 = E.PLANE_CONVERSIONS
            E.enable_event_log(True)
            with torch.inference_mode():
                out[flag] = net(x)
            torch.cuda.synchronize()
            tags[flag] = set(E.event_log_summary())
            E.enable_event_log(False)
            conv[flag] = E.PLANE_CONVERSIONS - c0
        finally:
            E.PREDICT_SKIP_PLANES = True
    assert any(t.startswith("gemm_planes_bf16<convT") for t in tags[True]), tags[True]
    assert not any(t.startswith("gemm_planes_bf16<convT") for t in tags[False])
    assert not any(t.startswith("conv_fwd_bf16<1,4x1,direct,d2s>") for t in tags[True]) or conv[True] <= conv[False]     # (deep, narrow levels may keep the split-K form)
    dd = float((out[True] - out[False]).abs().max())
    record_margin(f"predict_planes/{prec}/vs_fp32_skips", dd, 2e-2 if prec == "bf16" else 3e-3)
    assert dd <= (2e-2 if prec == "bf16" else 3e-3), dd
    assert torch.isfinite(out[True]).all()


SPECTRAL = [("net_spectral_f50", 1242, (2, 22, 9, 14), lambda H: H.SpectralUNET(22, 1, 50)),
            ("net_spectral1650_small", 1250, (2, 238, 16, 24), lambda H: H.SpectralUNET(238, 1, 1650)),
            ("net_spectral_nobn_f50", 1242, (2, 22, 9, 14), lambda H: H.SpectralUNET(22, 1, 50, bnorm=False)),
            ("net_spectral_3class", 1252, (2, 10, 7, 9), lambda H: H.SpectralUNET(10, 3, 4)),
            ("net_spectral_f48", 1238, (2, 22, 12, 20), lambda H: H.SpectralUNET(22, 1, 48)),
            ("net_spectral_tiny", 1237, (3, 10, 7, 9), lambda H: H.SpectralUNET(10, 1, 4))]


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("name,xseed,xshape,mk", SPECTRAL, ids=[c[0] for c in SPECTRAL])
def test_spectral_predict_runs_on_the_plane_gemm(name, xseed, xshape, mk, prec):
    """SpectralUNET's predict forward in the 16-bit modes (round 5): the folded Linear -> BatchNorm1d -> ReLU layers on the plane GEMM
    with the skips concatenated on planes (models.py:105-115,139-143 under PLTrainer.py:530-532), not on the round-1 kernel.  Against
    the reference's eval-mode fixture (the mode's band) and against the previous form (same arithmetic, another summation order)."""
    import hyperpri_amd as H
    from hyperpri_amd import engine as E
    z = np.load(os.path.join(G, name + ".npz"))
    net = mk(H)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = net.to(DEV).train()
    x = _u(xseed, xshape).to(DEV)
    net(x)                                       # one training forward in fp32: the fixture's running statistics
    H.set_precision(net, prec).eval()
    out, tags = {}, {}
    for flag in (True, False):
        E.PREDICT_GEMM_PLANES = flag
        try:
            E.enable_event_log(True)
            with torch.inference_mode():
                out[flag] = net(x)
            torch.cuda.synchronize()
            tags[flag] = set(E.event_log_summary())
            E.enable_event_log(False)
        finally:
            E.PREDICT_GEMM_PLANES = True
    assert any(t.startswith("gemm_planes_bf16<1") for t in tags[True]) and not any(t.startswith("conv_fwd_bf16<1") for t in tags[True]), tags[True]
    assert any(t.startswith("conv_fwd_bf16<1") for t in tags[False]), tags[False]
    le = out[True].cpu().numpy()
    band = 0.1 if prec == "bf16" else 0.02
    d = float(np.abs(le - z["logits_eval"]).max())
    record_margin(f"predict_gemm/{prec}/{name}/fixture", d, band)
    assert d < band, d
    dd = float((out[True] - out[False]).abs().max())
    record_margin(f"predict_gemm/{prec}/{name}/vs_round1_kernel", dd, band)
    assert dd < band, dd
