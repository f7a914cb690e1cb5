"""gemm_f32v2.hip (round 4): the exact-fp32 GEMM behind the ConvTranspose2d forms (model_parts.py:63-64) and the plain row GEMM,
called through the C ABI against fp64 of the same operands: the three modes, ragged row counts (partial pixel tiles), K not a
multiple of 16, column counts that are not multiples of the 128-column block, channel-slice views, accumulate, error returns.
fp32 MFMA = fp32 fma chain: the only difference to fp64 is fp32 rounding (tolerance 2e-6 of the output scale x sqrt(K)).
Needs a real MI355X: ``-m gpu``."""
import ctypes

import pytest
import torch

from conftest import record_margin

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    from hyperpri_amd import _lib
    return _lib.load()


def P(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def rup(x, m):
    return (x + m - 1) // m * m


def _pack(lib, w, mode, K, ncols, cup, d1):
    ncols_pad = rup(ncols, 64)
    wp = torch.empty(lib.hpri_packed_weight_f32k16_floats(K, ncols_pad), device=DEV)
    assert lib.hpri_pack_weight_f32k16(P(w), P(wp), mode, K, ncols, ncols_pad, cup, d1, _st()) == 0
    return wp, ncols_pad


@pytest.mark.parametrize("shape", [(1, 1, 16, 8), (2, 300, 40, 72), (1, 513, 128, 256), (3, 77, 100, 130), (1, 4096, 1024, 64)])
@pytest.mark.parametrize("acc", [0, 1])
def test_row_gemm_vs_fp64(lib, shape, acc):
    N, HW, K, ncols = shape
    torch.manual_seed(1)
    kpad, xcs = rup(K, 16), rup(K, 16) + 8
    x = torch.zeros(N * HW, xcs, device=DEV)
    x[:, 4:4 + K] = torch.randn(N * HW, K, device=DEV)                    # a channel-slice view: offset 4, zero pad behind K
    w = torch.randn(ncols, K, device=DEV) * 0.1
    b = torch.randn(ncols, device=DEV)
    wp, ncols_pad = _pack(lib, w, 0, K, ncols, 0, K)
    ycw, ycs = rup(ncols, 4), rup(ncols, 4) + 12
    y0 = torch.randn(N * HW * ycs + 16, device=DEV)                      # the output view starts 8 floats into the buffer
    y = y0.clone()
    rc = lib.hpri_gemm_f32v2(P(x), xcs, 4, P(wp), P(b), ctypes.c_void_p(y.data_ptr() + 32), ycs, 0, N, HW, kpad, ncols, ncols_pad, ycw, acc, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    ref = x[:, 4:4 + K].double() @ w.double().t() + b.double()
    view = lambda t, c0, cw: torch.as_strided(t, (N * HW, cw), (ycs, 1), 8 + c0)
    want = ref + (view(y0, 0, ncols).double() if acc else 0)
    err = float((view(y, 0, ncols).double() - want).abs().max())
    tol = 2e-6 * float(want.abs().max()) * max(1.0, K ** 0.5)
    record_margin(f"gemm_f32v2/mode0/{N}x{HW}x{K}x{ncols}/acc{acc}", err, tol)
    assert err <= tol, (err, tol)
    if ycw > ncols:
        assert bool((view(y, ncols, ycw - ncols) == 0).all())            # written pad columns hold zeros
    assert torch.equal(view(y, ycw, ycs - ycw), view(y0, ycw, ycs - ycw))    # nothing outside the written columns was touched
    assert torch.equal(y[:8], y0[:8])


@pytest.mark.parametrize("geom", [(2, 5, 7, 32, 16, 0, 0), (1, 19, 30, 64, 32, 0, 1), (2, 38, 60, 256, 128, 0, 0), (1, 8, 8, 16, 4, 1, 2)])
def test_transposed_convolution_forward_and_data_gradient_vs_fp64(lib, geom):
    """y = ConvTranspose2d(Cin, Cup, 2, 2)(x) placed at (py0, px0) inside an H2 x W2 canvas (F.pad), and dL/dx from dL/dy."""
    N, H, W, Cin, Cup, py0, px0 = geom
    torch.manual_seed(2)
    H2, W2 = 2 * H + py0 + (1 if py0 else 0), 2 * W + px0 + (1 if px0 else 0)
    xcs = rup(Cin, 16)
    x = torch.zeros(N * H * W, xcs, device=DEV)
    x[:, :Cin] = torch.randn(N * H * W, Cin, device=DEV)
    wt = torch.randn(Cin, Cup, 2, 2, device=DEV) * 0.1
    b = torch.randn(Cup, device=DEV)
    ref = torch.nn.functional.conv_transpose2d(x[:, :Cin].view(N, H, W, Cin).permute(0, 3, 1, 2).double(), wt.double(), b.double(), stride=2)
    # ---- forward
    wp, ncols_pad = _pack(lib, wt, 2, Cin, 4 * Cup, Cup, 0)
    ycs = Cup + 8
    y = torch.full((N * H2 * W2, ycs), 7.0, device=DEV)
    rc = lib.hpri_convt_fwd_f32v2(P(x), xcs, 0, P(wp), P(b), P(y), ycs, 4, N, H, W, rup(Cin, 16), Cup, ncols_pad, H2, W2, py0, px0, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    got = y.view(N, H2, W2, ycs)[:, py0:py0 + 2 * H, px0:px0 + 2 * W, 4:4 + Cup].permute(0, 3, 1, 2).double()
    err = float((got - ref).abs().max())
    tol = 2e-6 * float(ref.abs().max()) * max(1.0, Cin ** 0.5)
    record_margin(f"gemm_f32v2/convT_fwd/{geom}", err, tol)
    assert err <= tol, (err, tol)
    yy = y.view(N, H2, W2, ycs).clone()
    yy[:, py0:py0 + 2 * H, px0:px0 + 2 * W, 4:4 + Cup] = 7.0
    assert bool((yy == 7.0).all())                                  # pad ring and neighbouring channels untouched
    # ---- data gradient (Cup % 16 == 0 only)
    if Cup % 16:
        return
    dy = torch.zeros(N * H2 * W2, Cup + 16, device=DEV)
    dy[:, 16:16 + Cup] = torch.randn(N * H2 * W2, Cup, device=DEV)
    g = dy[:, 16:16 + Cup].view(N, H2, W2, Cup)[:, py0:py0 + 2 * H, px0:px0 + 2 * W].permute(0, 3, 1, 2).double()
    gref = torch.nn.functional.conv2d(g, wt.double().permute(0, 1, 2, 3), stride=2)               # dx[ci] = sum_{co,tap} dy * w[ci][co][tap]
    wpd, cin_pad = _pack(lib, wt, 3, 4 * Cup, Cin, Cup, 0)
    for acc in (0, 1):
        dx0 = torch.randn(N * H * W, rup(Cin, 4), device=DEV)
        dx = dx0.clone()
        rc = lib.hpri_convt_dgrad_f32v2(P(dy), Cup + 16, 16, P(wpd), P(dx), rup(Cin, 4), 0, N, H, W, Cup, Cin, cin_pad, rup(Cin, 4), H2, W2, py0, px0, acc, _st())
        assert rc == 0, lib.hpri_last_error()
        torch.cuda.synchronize()
        want = gref.permute(0, 2, 3, 1).reshape(N * H * W, Cin) + (dx0[:, :Cin].double() if acc else 0)
        err = float((dx[:, :Cin].double() - want).abs().max())
        tol = 2e-6 * float(want.abs().max()) * max(1.0, (4 * Cup) ** 0.5)
        record_margin(f"gemm_f32v2/convT_dgrad/{geom}/acc{acc}", err, tol)
        assert err <= tol, (err, tol)


def test_bad_arguments_are_error_returns(lib):
    z = torch.zeros(4096, device=DEV)
    assert lib.hpri_gemm_f32v2(P(z), 16, 0, P(z), P(None), P(z), 16, 0, 1, 4, 24, 8, 64, 8, 0, _st()) != 0        # K_pad not a multiple of 16
    assert lib.hpri_gemm_f32v2(P(z), 18, 0, P(z), P(None), P(z), 16, 0, 1, 4, 16, 8, 64, 8, 0, _st()) != 0        # row stride not a multiple of 4
    assert lib.hpri_convt_dgrad_f32v2(P(z), 8, 0, P(z), P(z), 16, 0, 1, 2, 2, 8, 16, 64, 16, 4, 4, 0, 0, 0, _st()) != 0   # Cup % 16
    assert lib.hpri_convt_fwd_f32v2(P(z), 16, 0, P(z), P(None), P(z), 8, 0, 1, 2, 2, 16, 8, 64, 3, 4, 0, 0, _st()) != 0   # canvas too small
