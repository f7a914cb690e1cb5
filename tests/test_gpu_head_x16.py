"""The 1x1 output layer (model_parts.py:96; models.py:103,143) over bf16 rows and over many channels (round 4), through the C ABI
against fp64 of the same operands: hpri_outconv_fwd_x16 / hpri_outconv_bwd_x16 (the head of the bf16 mode reads plane 0 of its
producer's plane buffer), the wide one-class kernel on fp32 rows (SpectralUNET's Linear(3300, 1)), the channel-block form of the data
gradient (any channel count, accumulate, channel-slice destination), with and without the loss inside the kernels, K = 1 and K = 3.
Tolerances: fp32 sums of C products (1e-5 x sqrt(C) of the output scale); the bf16 rows are exact inputs (the reference sees the
same rounded values).  Needs a real MI355X: ``-m gpu``."""
import ctypes

import pytest
import torch

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


def _source(N, HW, C, bf16, coff):
    """(buffer, cs, values as fp64 (N*HW, C)): rows of C channels at channel offset ``coff`` of a wider buffer, zero pad behind."""
    m = 32 if bf16 else 8
    cs = rup(C, m) + coff + (m if coff else 0)
    x = torch.zeros(N * HW, cs, device=DEV)
    x[:, coff:coff + C] = torch.randn(N * HW, C, device=DEV)
    if bf16:
        x = x.to(torch.bfloat16)
    return x, cs, x[:, coff:coff + C].double()


CASES = [  # N, HW, C, K, bf16 rows, channel offset
    (2, 1000, 64, 1, True, 0), (1, 333, 64, 1, True, 32), (1, 517, 3314, 1, True, 0), (2, 129, 1650, 1, True, 1664),
    (1, 517, 3300, 1, False, 0), (2, 300, 1650, 1, False, 8), (1, 200, 260, 1, False, 0), (2, 1000, 64, 1, False, 0),
    (1, 211, 40, 3, True, 0), (1, 211, 330, 3, False, 4), (1, 50, 6, 1, False, 0), (1, 50, 6, 1, True, 0),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("bce", [False, True])
def test_head_forward(lib, case, bce):
    N, HW, C, K, bf16, coff = case
    torch.manual_seed(3)
    x, cs, xr = _source(N, HW, C, bf16, coff)
    w = torch.randn(K, C, device=DEV) / C ** 0.5
    b = torch.randn(K, device=DEV)
    y = torch.full((N, K, HW), 7.0, device=DEV)
    tgt = (torch.rand(N, K, HW, device=DEV) > 0.7).float()
    nblk = lib.hpri_outconv_fwd_bce_blocks(N, HW)
    part = torch.zeros(nblk, dtype=torch.float64, device=DEV)
    if bf16:
        rc = lib.hpri_outconv_fwd_x16(P(x), cs, coff, P(w), P(b), P(y), P(tgt) if bce else P(None), P(part) if bce else P(None),
                                      nblk if bce else 0, N, HW, C, K, _st())
    elif bce:
        rc = lib.hpri_outconv_fwd_bce(P(x), cs, coff, P(w), P(b), P(y), P(tgt), P(part), nblk, N, HW, C, K, _st())
    else:
        rc = lib.hpri_outconv_fwd(P(x), cs, coff, P(w), P(b), P(y), N, HW, C, K, _st())
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    ref = (xr @ w.double().t() + b.double()).view(N, HW, K).permute(0, 2, 1)
    assert (y.double() - ref).abs().max().item() <= 1e-5 * C ** 0.5 * max(1.0, ref.abs().max().item())
    if bce:
        loss = torch.empty((), device=DEV)
        assert lib.hpri_bce_finish(P(part), nblk, y.numel(), P(loss), _st()) == 0
        want = torch.nn.functional.binary_cross_entropy_with_logits(y.double(), tgt.double())
        assert abs(loss.item() - want.item()) <= 1e-6 * max(1.0, abs(want.item()))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("bce", [False, True])
@pytest.mark.parametrize("acc", [0, 1])
def test_head_backward(lib, case, bce, acc):
    N, HW, C, K, bf16, coff = case
    torch.manual_seed(4)
    x, cs, xr = _source(N, HW, C, bf16, coff)
    w = torch.randn(K, C, device=DEV) / C ** 0.5
    tgt = (torch.rand(N, K, HW, device=DEV) > 0.7).float()
    logits = torch.randn(N, K, HW, device=DEV)
    gs = torch.tensor([0.5], device=DEV)
    if bce:
        dy = (torch.sigmoid(logits.double()) - tgt.double()) * 0.5 / logits.numel()
        src = logits
    else:
        src = torch.randn(N, K, HW, device=DEV)
        dy = src.double()
    dcw, doff = rup(C, 4), 4
    dcs = dcw + 12
    dx0 = torch.randn(N * HW, dcs, device=DEV)
    dx = dx0.clone()
    dw0, db0 = torch.randn(K, C, device=DEV), torch.randn(K, device=DEV)
    dw, db = dw0.clone(), db0.clone()
    nblk, cpart = ctypes.c_int(), ctypes.c_int()
    lib.hpri_outconv_bwd_plan(N, HW, C, K, ctypes.byref(nblk), ctypes.byref(cpart))
    ws = torch.empty(nblk.value * K * 2 * cpart.value, device=DEV)
    common = (P(w), P(dx), dcs, doff, dcw, acc, P(dw), P(db), acc, P(ws), ws.numel(), N, HW, C, K, _st())
    if bf16:
        rc = lib.hpri_outconv_bwd_x16(P(src), P(tgt) if bce else P(None), P(gs) if bce else P(None), P(x), cs, coff, P(w), P(dx), 0, *common[2:])
    elif bce:
        rc = lib.hpri_outconv_bwd_bce(P(src), P(tgt), P(gs), P(x), cs, coff, *common)
    else:
        rc = lib.hpri_outconv_bwd(P(src), P(x), cs, coff, *common)
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    dyr = dy.permute(0, 2, 1).reshape(N * HW, K)                       # (pixels, K)
    scale = dy.abs().max().item()
    want_dx = dyr @ w.double() + (dx0[:, doff:doff + C].double() if acc else 0)
    assert (dx[:, doff:doff + C].double() - want_dx).abs().max().item() <= 1e-5 * max(scale, want_dx.abs().max().item())
    assert torch.equal(dx[:, :doff], dx0[:, :doff]) and torch.equal(dx[:, doff + dcw:], dx0[:, doff + dcw:])     # the neighbours of the slice
    if dcw > C and not acc:
        assert float(dx[:, doff + C:doff + dcw].abs().max()) == 0.0                                               # pad channels: zeros
    want_dw = dyr.t() @ xr + (dw0.double() if acc else 0)
    want_db = dyr.sum(0) + (db0.double() if acc else 0)
    tol = 2e-5 * (N * HW) ** 0.5 * scale * max(1.0, xr.abs().max().item())
    assert (dw.double() - want_dw).abs().max().item() <= tol + 1e-6 * want_dw.abs().max().item()
    assert (db.double() - want_db).abs().max().item() <= tol + 1e-6 * want_db.abs().max().item()


@pytest.mark.parametrize("bce", [False, True])
@pytest.mark.parametrize("acc", [0, 1])
def test_head_backward_writes_bf16_rows(lib, bce, acc):
    """dx_bf16: the input gradient of the one-class head as bf16 rows == the fp32 result rounded once (accumulate: read, add, round)."""
    N, HW, C, K = 2, 777, 64, 1
    torch.manual_seed(5)
    x, cs, xr = _source(N, HW, C, True, 0)
    w = torch.randn(K, C, device=DEV) / C ** 0.5
    tgt = (torch.rand(N, K, HW, device=DEV) > 0.7).float()
    src = torch.randn(N, K, HW, device=DEV)
    gs = torch.tensor([0.5], device=DEV)
    nblk, cpart = ctypes.c_int(), ctypes.c_int()
    lib.hpri_outconv_bwd_plan(N, HW, C, K, ctypes.byref(nblk), ctypes.byref(cpart))
    ws = torch.empty(nblk.value * K * 2 * cpart.value, device=DEV)
    d16 = torch.randn(N * HW, C, device=DEV).to(torch.bfloat16)
    want = d16.float()
    dw, db = torch.zeros(K, C, device=DEV), torch.zeros(K, device=DEV)
    args = (P(src), P(tgt) if bce else P(None), P(gs) if bce else P(None), P(x), cs, 0, P(w))
    tail = (C, 0, C, acc, P(dw), P(db), 0, P(ws), ws.numel(), N, HW, C, K, _st())
    assert lib.hpri_outconv_bwd_x16(*args, P(want), 0, *tail) == 0, lib.hpri_last_error()
    got = d16.clone()
    assert lib.hpri_outconv_bwd_x16(*args, P(got), 1, *tail) == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    assert torch.equal(got, want.to(torch.bfloat16))
    w3 = torch.randn(3, C, device=DEV)
    assert lib.hpri_outconv_bwd_x16(P(torch.randn(N, 3, HW, device=DEV)), P(None), P(None), P(x), cs, 0, P(w3), P(got), 1, C, 0, C, 0,
                                    P(torch.zeros(3, C, device=DEV)), P(None), 0, P(torch.empty(1 << 20, device=DEV)), 1 << 20, N, HW, C, 3, _st()) != 0


def test_head_rejects_bad_layouts(lib):
    x = torch.zeros(64, 64, device=DEV, dtype=torch.bfloat16)
    w, y = torch.zeros(1, 64, device=DEV), torch.zeros(1, 1, 64, device=DEV)
    assert lib.hpri_outconv_fwd_x16(P(x), 62, 0, P(w), P(None), P(y), P(None), P(None), 0, 1, 64, 64, 1, _st()) != 0      # stride not a multiple of 4
    assert lib.hpri_outconv_fwd_x16(P(x), 64, 0, P(w), P(None), P(y), P(y), P(None), 0, 1, 64, 64, 1, _st()) != 0       # target without partials
    assert lib.hpri_outconv_fwd_x16(P(None), 64, 0, P(w), P(None), P(y), P(None), P(None), 0, 1, 64, 64, 1, _st()) != 0
