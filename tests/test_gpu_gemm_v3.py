"""gemm_bf16v3.hip through the C ABI (include/hyperpri_hip.h) against fp64 references on the bf16-rounded operands: the
plane-fed 1x1 forms of the bf16 precision mode -- nn.Linear / Conv2d(k=1) forward and data gradient (reference
models.py:105-115,143), ConvTranspose2d(k=2,s=2) forward (depth-to-space epilogue) and data gradient (space-to-depth gather;
model_parts.py:63-64).  Ragged row counts, several images (tiles never straddle one), channel-slice views, accumulate / ReLU,
fp32 and bf16 outputs, per-tile BatchNorm records.  Needs a real MI355X: ``-m gpu``."""
import ctypes

import pytest
import torch

from conftest import record_margin

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rup(x, m):
    return (x + m - 1) // m * m


def P(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


@pytest.fixture(scope="module")
def lib():
    from hyperpri_amd import _lib
    return _lib.load()


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _planes(x, cs):
    pl = torch.zeros(x.shape[0], cs, dtype=torch.bfloat16, device=DEV)
    pl[:, :x.shape[1]] = x.to(torch.bfloat16)
    return pl


@pytest.mark.parametrize("shape", [(1, 300, 40, 24), (2, 777, 238, 150), (1, 1024, 96, 128), (3, 257, 64, 330), (1, 5, 8, 4), (2, 4100, 1650, 200)])
def test_linear_forward_statistics_and_outputs(lib, shape):
    """y = x W^T + b with per-tile BatchNorm records; the fp32 view sits inside a wider buffer, the bf16 view is written by the
    same launch; then ReLU + accumulate into what is there."""
    N, HW, K, C = shape
    torch.manual_seed(3)
    kp, cp, cw = rup(K, 32), rup(C, 64), rup(C, 4)
    x = torch.randn(N * HW, K, device=DEV)
    xp = _planes(x, kp + 8)[:, :]                                          # row stride kp + 8 (a view wider than K_pad)
    w = torch.randn(C, K, device=DEV) * 0.1
    b = torch.randn(C, device=DEV)
    wp = torch.empty((kp // 32) * cp * 32, dtype=torch.bfloat16, device=DEV)
    assert lib.hpri_pack_weight_bf16(P(w), P(wp), 0, K, C, cp, 1, K, 0, 0, _st()) == 0, lib.hpri_last_error()
    tl = ctypes.c_int()
    assert lib.hpri_gemm_bf16v3_plan(N, HW, ctypes.byref(tl)) == 0
    assert tl.value == N * ((HW + 255) // 256)
    ycs, yoff = cw + 12, 4
    ybuf = torch.full((N * HW, ycs), 3.0, device=DEV)
    y16 = torch.full((N * HW, cw), 5.0, dtype=torch.bfloat16, device=DEV)
    stats = torch.full((tl.value * cp * 4,), float("nan"), device=DEV)
    rc = lib.hpri_gemm_bf16v3(P(xp), kp + 8, 0, P(wp), P(b), P(ybuf), ycs, yoff, P(y16), cw, 0, P(stats), cp, N, HW, kp, C, cp, cw, 0, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    ref = xp[:, :K].double().cpu() @ w.to(torch.bfloat16).double().cpu().T + b.double().cpu()
    sc = max(1.0, float(ref.abs().max()))
    got = ybuf.double().cpu()
    err = float((got[:, yoff:yoff + C] - ref).abs().max())
    record_margin(f"gemm_bf16v3/fwd/{N}x{HW}x{K}x{C}", err, 2e-5 * sc)
    assert err < 2e-5 * sc, (shape, err)
    assert float(got[:, :yoff].sub(3.0).abs().max()) == 0.0 and float(got[:, yoff + cw:].sub(3.0).abs().max()) == 0.0
    if cw > C:
        assert float(got[:, yoff + C:yoff + cw].abs().max()) == 0.0          # pad columns: exact zeros
    assert torch.equal(y16[:, :C], ybuf[:, yoff:yoff + C].to(torch.bfloat16))   # the bf16 view holds the same values, rounded once
    # statistics: per image (tiles of one image are whole records)
    s = stats.view(N, tl.value // N, cp, 4).double().cpu()
    n = s[..., 2]
    assert torch.all(n[:, :, :C].sum(1) == HW)
    mean = (s[..., 0] * n).sum(1) / n.sum(1)
    m2 = (s[..., 1] + n * (s[..., 0] - mean[:, None]) ** 2).sum(1)
    r = ref.view(N, HW, C)
    assert float((mean[:, :C] - r.mean(1)).abs().max()) < 1e-4 * sc
    assert float((m2[:, :C] / HW - r.var(1, unbiased=False)).abs().max()) < 1e-4 * sc * sc
    # ReLU + accumulate, no statistics, no bf16 view
    prior = ybuf.clone()
    rc = lib.hpri_gemm_bf16v3(P(xp), kp + 8, 0, P(wp), P(b), P(ybuf), ycs, yoff, P(None), 0, 0, P(None), 0, N, HW, kp, C, cp, cw, 3, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    want = prior.double().cpu()[:, yoff:yoff + C] + torch.relu(ref)
    assert float((ybuf.double().cpu()[:, yoff:yoff + C] - want).abs().max()) < 4e-5 * sc
    # boundary errors
    assert lib.hpri_gemm_bf16v3(P(xp), kp + 8, 0, P(wp), P(b), P(None), 0, 0, P(None), 0, 0, P(None), 0, N, HW, kp, C, cp, cw, 0, _st()) != 0
    assert lib.hpri_gemm_bf16v3(P(xp), kp - 8, 0, P(wp), P(b), P(ybuf), ycs, yoff, P(None), 0, 0, P(None), 0, N, HW, kp, C, cp, cw, 0, _st()) != 0
    assert lib.hpri_gemm_bf16v3(P(xp), kp + 8, 0, P(wp), P(b), P(ybuf), ycs, yoff + 1, P(None), 0, 0, P(None), 0, N, HW, kp, C, cp, cw, 0, _st()) != 0
    assert lib.hpri_gemm_bf16v3(P(xp), kp + 8, 0, P(wp), P(b), P(ybuf), ycs, yoff, P(None), 0, 0, P(stats), cp, N, HW, kp, C, cp, cw, 1, _st()) != 0


@pytest.mark.parametrize("shape", [(2, 500, 150, 238), (1, 1300, 64, 96)])
def test_linear_data_gradient(lib, shape):
    """dx = dy W (pack mode 1: K = out features, columns = in features), accumulating."""
    N, HW, Cout, Cin = shape
    torch.manual_seed(5)
    kp, cp, cw = rup(Cout, 32), rup(Cin, 64), rup(Cin, 4)
    dy = torch.randn(N * HW, Cout, device=DEV)
    dyp = _planes(dy, kp)
    w = torch.randn(Cout, Cin, device=DEV) * 0.1
    wp = torch.empty((kp // 32) * cp * 32, dtype=torch.bfloat16, device=DEV)
    assert lib.hpri_pack_weight_bf16(P(w), P(wp), 1, Cout, Cin, cp, 1, Cin, 0, 0, _st()) == 0, lib.hpri_last_error()
    prior = torch.randn(N * HW, cw, device=DEV)
    dx = prior.clone()
    rc = lib.hpri_gemm_bf16v3(P(dyp), kp, 0, P(wp), P(None), P(dx), cw, 0, P(None), 0, 0, P(None), 0, N, HW, kp, Cin, cp, cw, 1, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    ref = dyp[:, :Cout].double().cpu() @ w.to(torch.bfloat16).double().cpu()
    sc = max(1.0, float(ref.abs().max()))
    err = float((dx.double().cpu()[:, :Cin] - prior.double().cpu()[:, :Cin] - ref).abs().max())
    record_margin(f"gemm_bf16v3/dgrad/{N}x{HW}x{Cout}x{Cin}", err, 2e-5 * sc)
    assert err < 2e-5 * sc, (shape, err)
    # round 4: the same product ADDED into bf16 rows (a gradient with two producers stored as bf16) == bf16(fp32 sum of the old rows
    # and the fp32 product); a bias, ReLU or the transposed-convolution forms refuse that combination
    fresh = torch.zeros(N * HW, cw, device=DEV)
    assert lib.hpri_gemm_bf16v3(P(dyp), kp, 0, P(wp), P(None), P(fresh), cw, 0, P(None), 0, 0, P(None), 0, N, HW, kp, Cin, cp, cw, 0, _st()) == 0
    old16 = torch.randn(N * HW, cw + 8, device=DEV).to(torch.bfloat16)
    got16 = old16.clone()
    rc = lib.hpri_gemm_bf16v3(P(dyp), kp, 0, P(wp), P(None), P(None), 0, 0, P(got16), cw + 8, 0, P(None), 0, N, HW, kp, Cin, cp, cw, 1, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    assert torch.equal(got16[:, :cw], (old16[:, :cw].float() + fresh).to(torch.bfloat16)) and torch.equal(got16[:, cw:], old16[:, cw:])
    b = torch.zeros(Cin, device=DEV)
    assert lib.hpri_gemm_bf16v3(P(dyp), kp, 0, P(wp), P(b), P(None), 0, 0, P(got16), cw + 8, 0, P(None), 0, N, HW, kp, Cin, cp, cw, 1, _st()) != 0
    assert lib.hpri_gemm_bf16v3(P(dyp), kp, 0, P(wp), P(None), P(None), 0, 0, P(got16), cw + 8, 0, P(None), 0, N, HW, kp, Cin, cp, cw, 3, _st()) != 0


@pytest.mark.parametrize("shape", [(2, 9, 15, 128, 64, 0, 0), (1, 19, 30, 64, 32, 1, 1), (2, 5, 7, 96, 48, 0, 1), (1, 38, 60, 256, 128, 0, 0)])
def test_transposed_convolution_forward_and_data_gradient(lib, shape):
    """ConvTranspose2d(Cin, Cup, 2, 2): forward into a channel-slice view of a padded [N, H2, W2] buffer (fp32 and bf16 at once),
    then the data gradient gathered from bf16 planes of that geometry."""
    N, H, W, Cin, Cup, dY, dX = shape
    torch.manual_seed(9)
    H2, W2 = 2 * H + dY, 2 * W + dX
    py0, px0 = dY // 2, dX // 2
    kp, ncp = rup(Cin, 32), rup(4 * Cup, 64)
    x = torch.randn(N * H * W, Cin, device=DEV)
    xp = _planes(x, kp)
    wt = torch.randn(Cin, Cup, 2, 2, device=DEV) * 0.1
    b = torch.randn(Cup, device=DEV)
    wp = torch.empty((kp // 32) * ncp * 32, dtype=torch.bfloat16, device=DEV)
    assert lib.hpri_pack_weight_bf16(P(wt), P(wp), 2, Cin, 4 * Cup, ncp, 1, Cup, Cup, 0, _st()) == 0, lib.hpri_last_error()
    cskip = 8
    ycs = cskip + Cup + 4
    y = torch.full((N * H2 * W2, ycs), 2.0, device=DEV)
    y16 = torch.full((N * H2 * W2, ycs), 2.0, dtype=torch.bfloat16, device=DEV)
    rc = lib.hpri_convt_fwd_bf16v3(P(xp), kp, 0, P(wp), P(b), P(y), ycs, cskip, P(y16), ycs, cskip, N, H, W, kp, Cup, ncp, H2, W2, py0, px0, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    xr = xp[:, :Cin].double().cpu().view(N, H, W, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv_transpose2d(xr, wt.to(torch.bfloat16).double().cpu(), b.double().cpu(), stride=2)      # [N, Cup, 2H, 2W]
    full = torch.full((N, H2, W2, Cup), 2.0, dtype=torch.float64)
    full[:, py0:py0 + 2 * H, px0:px0 + 2 * W] = ref.permute(0, 2, 3, 1)
    got = y.double().cpu().view(N, H2, W2, ycs)
    sc = max(1.0, float(ref.abs().max()))
    err = float((got[..., cskip:cskip + Cup] - full).abs().max())
    record_margin(f"gemm_bf16v3/convt_fwd/{N}x{H}x{W}x{Cin}x{Cup}", err, 2e-5 * sc)
    assert err < 2e-5 * sc, (shape, err)                                    # (the pad ring and the other channels keep their 2.0)
    assert float(got[..., :cskip].sub(2.0).abs().max()) == 0.0 and float(got[..., cskip + Cup:].sub(2.0).abs().max()) == 0.0
    assert torch.equal(y16.view(N, H2, W2, ycs)[..., cskip:cskip + Cup], y.view(N, H2, W2, ycs)[..., cskip:cskip + Cup].to(torch.bfloat16))
    if Cup % 32:
        return
    # data gradient: dy planes [N, H2, W2, Cup] -> dx [N, H, W, Cin]
    dy = torch.randn(N, H2, W2, Cup, device=DEV)
    dcs = rup(Cup, 32) + 32
    dyp = torch.zeros(N * H2 * W2, dcs, dtype=torch.bfloat16, device=DEV)
    dyp[:, 32:32 + Cup] = dy.view(-1, Cup).to(torch.bfloat16)
    cinp, dcw = rup(Cin, 64), rup(Cin, 4)
    wpd = torch.empty((4 * Cup // 32) * cinp * 32, dtype=torch.bfloat16, device=DEV)
    assert lib.hpri_pack_weight_bf16(P(wt), P(wpd), 3, 4 * Cup, Cin, cinp, 1, Cup, Cup, 0, _st()) == 0, lib.hpri_last_error()
    prior = torch.randn(N * H * W, dcw, device=DEV)
    dx = prior.clone()
    rc = lib.hpri_convt_dgrad_bf16v3(P(dyp), dcs, 32, P(wpd), P(dx), dcw, 0, N, H, W, Cup, Cin, cinp, dcw, H2, W2, py0, px0, 1, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    dyr = dyp[:, 32:32 + Cup].double().cpu().view(N, H2, W2, Cup)[:, py0:py0 + 2 * H, px0:px0 + 2 * W].permute(0, 3, 1, 2)
    refd = torch.nn.functional.conv2d(dyr, wt.to(torch.bfloat16).double().cpu(), stride=2)          # [N, Cin, H, W]: the adjoint
    refd = refd.permute(0, 2, 3, 1).reshape(-1, Cin)
    scd = max(1.0, float(refd.abs().max()))
    errd = float((dx.double().cpu()[:, :Cin] - prior.double().cpu()[:, :Cin] - refd).abs().max())
    record_margin(f"gemm_bf16v3/convt_dgrad/{N}x{H}x{W}x{Cin}x{Cup}", errd, 2e-5 * scd)
    assert errd < 2e-5 * scd, (shape, errd)
    # the same written as bf16 rows (round 4): the fp32 result, rounded once
    dxf = torch.zeros(N * H * W, dcw, device=DEV)
    assert lib.hpri_convt_dgrad_bf16v3(P(dyp), dcs, 32, P(wpd), P(dxf), dcw, 0, N, H, W, Cup, Cin, cinp, dcw, H2, W2, py0, px0, 0, _st()) == 0
    d16 = torch.full((N * H * W, dcw + 8), 3.0, dtype=torch.bfloat16, device=DEV)
    rc = lib.hpri_convt_dgrad_bf16v3_y16(P(dyp), dcs, 32, P(wpd), P(d16), dcw + 8, 0, N, H, W, Cup, Cin, cinp, dcw, H2, W2, py0, px0, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    assert torch.equal(d16[:, :dcw], dxf.to(torch.bfloat16)) and float(d16[:, dcw:].float().sub(3.0).abs().max()) == 0.0
    assert lib.hpri_convt_dgrad_bf16v3_y16(P(dyp), dcs, 32, P(wpd), P(None), dcw + 8, 0, N, H, W, Cup, Cin, cinp, dcw, H2, W2, py0, px0, _st()) != 0


@pytest.mark.parametrize("shape", [(300, 40, 24), (5000, 238, 150), (70000, 330, 520), (33, 8, 8), (20000, 1650, 264)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_linear_weight_gradient_from_planes(lib, shape, accumulate):
    """wgrad_bf16v3.hip: dW[n][c] = sum_p dY[p][n] X[p][c] from bf16 planes (channel-slice views of wider plane buffers, valid
    widths below the tile widths, ragged pixel counts), slabs summed by hpri_wgrad_reduce_ex."""
    Ppx, Cin, Cout = shape
    torch.manual_seed(13)
    xcs, ycs = rup(Cin, 32) + 32, rup(Cout, 32) + 8
    xoff, yoff = 32, 8
    x = torch.randn(Ppx, Cin, device=DEV)
    dy = torch.randn(Ppx, Cout, device=DEV)
    xp = torch.full((Ppx, xcs), 9.0, dtype=torch.bfloat16, device=DEV)          # channels outside the valid width hold junk on purpose
    yp = torch.full((Ppx, ycs), 9.0, dtype=torch.bfloat16, device=DEV)
    xv, yv = rup(Cin, 8), rup(Cout, 8)
    xp[:, xoff:xoff + xv] = 0
    yp[:, yoff:yoff + yv] = 0
    xp[:, xoff:xoff + Cin] = x.to(torch.bfloat16)
    yp[:, yoff:yoff + Cout] = dy.to(torch.bfloat16)
    sp, cr, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.hpri_wgrad1x1_bf16v3_plan(Ppx, rup(Cin, 32), rup(Cout, 64), ctypes.byref(sp), ctypes.byref(cr), ctypes.byref(nr)) == 0
    assert cr.value % 128 == 0 and nr.value % 256 == 0 and cr.value >= Cin and nr.value >= Cout and sp.value >= 1
    ws = torch.full((sp.value * cr.value * nr.value,), float("nan"), device=DEV)
    rc = lib.hpri_wgrad1x1_bf16v3(P(xp), xcs, xoff, xv, P(yp), ycs, yoff, yv, P(ws), ws.numel(), Ppx, rup(Cin, 32), rup(Cout, 64), _st())
    assert rc == 0, lib.hpri_last_error()
    prior = torch.randn(Cout, Cin, device=DEV)
    dw = prior.clone()
    rc = lib.hpri_wgrad_reduce_ex(P(ws), P(dw), sp.value, cr.value, nr.value, Cin, Cout, 1, 0, 0, accumulate, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(ws).all()
    ref = yp[:, yoff:yoff + Cout].double().cpu().T @ xp[:, xoff:xoff + Cin].double().cpu()
    if accumulate:
        ref = ref + prior.double().cpu()
    sc = max(1.0, float(ref.abs().max()))
    err = float((dw.double().cpu() - ref).abs().max())
    record_margin(f"wgrad1x1_bf16v3/{Ppx}x{Cin}x{Cout}/acc{accumulate}", err, 3e-5 * sc)
    assert err < 3e-5 * sc, (shape, err, sp.value)
    # boundary: unaligned valid width, workspace too small
    assert lib.hpri_wgrad1x1_bf16v3(P(xp), xcs, xoff, xv + 4, P(yp), ycs, yoff, yv, P(ws), ws.numel(), Ppx, rup(Cin, 32), rup(Cout, 64), _st()) != 0
    assert lib.hpri_wgrad1x1_bf16v3(P(xp), xcs, xoff, xv, P(yp), ycs, yoff, yv, P(ws), ws.numel() - 1, Ppx, rup(Cin, 32), rup(Cout, 64), _st()) != 0


@pytest.mark.parametrize("shape", [(2, 9, 15, 128, 64, 0, 0), (1, 19, 30, 96, 64, 1, 1), (2, 38, 60, 256, 128, 0, 0), (1, 76, 121, 200, 64, 0, 1)])
def test_transposed_convolution_weight_gradient_from_planes(lib, shape):
    """hpri_wgrad_convt_bf16v3: dW[ci][co][tap] = sum_p x[p][ci] * dy[up(p, tap)][co], x planes at low resolution, dy planes at
    high resolution (a channel-slice view, pad ring present in two cases), gathered by parity inside the DMA offsets."""
    N, H, W, Cin, Cup, dY, dX = shape
    torch.manual_seed(17)
    H2, W2 = 2 * H + dY, 2 * W + dX
    py0, px0 = dY // 2, dX // 2
    xcs = rup(Cin, 32)
    x = torch.randn(N * H * W, Cin, device=DEV)
    xp = _planes(x, xcs)
    dcs = Cup + 64
    dy = torch.randn(N, H2, W2, Cup, device=DEV)
    dyp = torch.full((N * H2 * W2, dcs), 7.0, dtype=torch.bfloat16, device=DEV)
    dyp[:, 64:64 + Cup] = dy.view(-1, Cup).to(torch.bfloat16)
    sp, cr, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.hpri_wgrad1x1_bf16v3_plan(N * H * W, xcs, 4 * Cup, ctypes.byref(sp), ctypes.byref(cr), ctypes.byref(nr)) == 0
    ws = torch.full((sp.value * cr.value * nr.value,), float("nan"), device=DEV)
    rc = lib.hpri_wgrad_convt_bf16v3(P(xp), xcs, 0, rup(Cin, 8), P(dyp), dcs, 64, P(ws), ws.numel(), N, H, W, xcs, Cup, H2, W2, py0, px0, _st())
    assert rc == 0, lib.hpri_last_error()
    dw = torch.zeros(Cin, Cup, 2, 2, device=DEV)
    rc = lib.hpri_wgrad_reduce_ex(P(ws), P(dw), sp.value, cr.value, nr.value, Cin, 4 * Cup, 1, 1, Cup, 0, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    xr = xp[:, :Cin].double().cpu().view(N, H, W, Cin)
    dyr = dyp[:, 64:64 + Cup].double().cpu().view(N, H2, W2, Cup)[:, py0:py0 + 2 * H, px0:px0 + 2 * W]
    ref = torch.empty(Cin, Cup, 2, 2, dtype=torch.float64)
    for a in range(2):
        for b in range(2):
            ref[:, :, a, b] = torch.einsum("nhwc,nhwk->ck", xr, dyr[:, a::2, b::2])
    sc = max(1.0, float(ref.abs().max()))
    err = float((dw.double().cpu() - ref).abs().max())
    record_margin(f"wgrad_convt_bf16v3/{N}x{H}x{W}x{Cin}x{Cup}", err, 3e-5 * sc)
    assert err < 3e-5 * sc, (shape, err, sp.value)
    assert lib.hpri_wgrad_convt_bf16v3(P(xp), xcs, 0, rup(Cin, 8), P(dyp), dcs, 64, P(ws), ws.numel(), N, H, W, xcs, Cup + 32, H2, W2, py0, px0, _st()) != 0


@pytest.mark.parametrize("shape", [(2, 36, 50, 64, 128, 64, 64), (1, 76, 121, 128, 256, 128, 128), (2, 19, 30, 32, 192, 128, 64)])
@pytest.mark.parametrize("only", [0, 1])
def test_bf16_plane_conv_v3_second_output(lib, shape, only):
    """hpri_conv_bf16v3_y2: the data gradient of the first convolution of a decoder stage also (only = 1: only) leaves the channel
    blocks of the upsampled half as bf16 rows; everything else is the plain launch bit for bit.  The per-tile statistics are the same
    records up to summation order (round 4: the plain launch sums them after its LDS transposition, this form straight from the
    accumulators): counts equal, means and M2 to 1e-5 of their scale."""
    N, H, W, K, Cols, c0, cw2 = shape
    torch.manual_seed(19)
    cs16, cols_pad, cw = rup(K, 32), rup(Cols, 64), rup(Cols, 8)
    npx = N * H * W
    dy = torch.randn(npx, K, device=DEV)
    planes = torch.zeros(npx, cs16, dtype=torch.bfloat16, device=DEV)
    planes[:, :K] = dy.to(torch.bfloat16)
    w = torch.randn(K, Cols, 3, 3, device=DEV) * 0.05
    wpd = torch.empty(((K + 31) // 32) * 9 * cols_pad * 32, dtype=torch.bfloat16, device=DEV)
    assert lib.hpri_pack_weight_bf16(P(w), P(wpd), 1, K, Cols, cols_pad, 9, Cols, 0, 0, _st()) == 0
    k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    lib.hpri_conv_bf16v3_plan(N, H, W, cs16, cols_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
    assert k.value == 1
    st1 = torch.empty(tl.value * cols_pad * 4, device=DEV)
    st2 = torch.empty_like(st1)
    g_plain = torch.empty(npx, cw, device=DEV)
    assert lib.hpri_conv_bf16v3(P(planes), 0, cs16, 0, P(wpd), P(None), P(g_plain), cw, 0, P(st1), N, H, W, cs16, Cols, cols_pad, cw, 0, 0,
                                P(None), 0, _st()) == 0
    g = torch.full((npx, cw), -77.0, device=DEV)
    y2cs = cw2 + 8
    y2 = torch.full((npx, y2cs), 5.0, dtype=torch.bfloat16, device=DEV)
    rc = lib.hpri_conv_bf16v3_y2(P(planes), cs16, 0, P(wpd), P(None), P(g), cw, 0, P(st2), N, H, W, cs16, Cols, cols_pad, cw, P(y2), y2cs, 4, c0,
                                 cw2, only, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    r1, r2 = st1.view(-1, 4), st2.view(-1, 4)
    assert torch.equal(r1[:, 2], r2[:, 2])
    for j in (0, 1):
        assert float((r1[:, j] - r2[:, j]).abs().max()) <= 1e-5 * max(1.0, float(r1[:, j].abs().max()))
    assert torch.equal(y2[:, 4:4 + cw2], g_plain[:, c0:c0 + cw2].to(torch.bfloat16))
    assert float(y2[:, :4].float().sub(5.0).abs().max()) == 0.0 and float(y2[:, 4 + cw2:].float().sub(5.0).abs().max()) == 0.0
    assert torch.equal(g[:, :c0], g_plain[:, :c0]) and torch.equal(g[:, c0 + cw2:], g_plain[:, c0 + cw2:])
    if only:
        assert float(g[:, c0:c0 + cw2].sub(-77.0).abs().max()) == 0.0          # untouched
    else:
        assert torch.equal(g[:, c0:c0 + cw2], g_plain[:, c0:c0 + cw2])
    assert lib.hpri_conv_bf16v3_y2(P(planes), cs16, 0, P(wpd), P(None), P(g), cw, 0, P(st2), N, H, W, cs16, Cols, cols_pad, cw, P(y2), y2cs, 4, c0 + 32,
                                   cw2, only, _st()) != 0
    if only and c0 + cw2 == Cols and c0 > 0:
        # flags bit 1 (round 4): the main output as COMPACT bf16 rows of the channels below the second output's range
        g16 = torch.full((npx, c0 + 8), 9.0, dtype=torch.bfloat16, device=DEV)
        y2b = torch.full_like(y2, 5.0)
        st3 = torch.empty_like(st1)
        rc = lib.hpri_conv_bf16v3_y2(P(planes), cs16, 0, P(wpd), P(None), P(g16), c0 + 8, 0, P(st3), N, H, W, cs16, Cols, cols_pad, c0, P(y2b), y2cs, 4,
                                     c0, cw2, 3, _st())
        assert rc == 0, lib.hpri_last_error()
        torch.cuda.synchronize()
        assert torch.equal(g16[:, :c0], g_plain[:, :c0].to(torch.bfloat16)) and float(g16[:, c0:].float().sub(9.0).abs().max()) == 0.0
        assert torch.equal(y2b, y2) and torch.equal(st3, st2)
        # ... which must reach Cout from below the range
        assert lib.hpri_conv_bf16v3_y2(P(planes), cs16, 0, P(wpd), P(None), P(g16), c0 + 8, 0, P(st3), N, H, W, cs16, Cols, cols_pad, c0 - 4,
                                       P(y2b), y2cs, 4, c0, cw2, 3, _st()) != 0
