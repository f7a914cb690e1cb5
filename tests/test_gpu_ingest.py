"""Ingest fast path (hyperpri_amd/ingest.py, csrc/ingest.hip): (H,W,B) cubes -> zero-copy network input.
Bit-exact data movement; the network's output must be bit-identical to the one it produces from the reference's
(N,1,B,H,W) layout.  Needs a real MI355X: ``-m gpu``."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hyperpri_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _u(seed, shape):
    return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())


@pytest.mark.parametrize("B,lo,hi", [(11, 2, 9), (8, 0, 8), (238, 0, 238), (299, 0, 238), (13, 12, 13)])
def test_from_hwb_slices_converts_and_pads(B, lo, hi):
    from hyperpri_amd.ingest import from_hwb
    cube = _u(61, (2, 9, 14, B))
    for dt in (torch.float32, torch.float16):
        x = from_hwb(cube.to(dt).to(DEV), lo, hi)
        want = cube.to(dt).float()[..., lo:hi].permute(0, 3, 1, 2).unsqueeze(1)      # dataset.py:267-270
        assert x.shape == want.shape and torch.equal(x.cpu(), want)
        cs = x.stride(4)
        assert cs % 8 == 0 and cs >= hi - lo and x.stride(2) == 1
        under = torch.as_strided(x, (2, 9, 14, cs), (9 * 14 * cs, 14 * cs, cs, 1))   # the padded buffer itself
        assert torch.count_nonzero(under[..., hi - lo:]) == 0


@pytest.mark.parametrize("direct", [True, False])
def test_stager_round_trip_and_slot_reuse(direct):
    from hyperpri_amd.ingest import CubeStager
    st = CubeStager(2, 12, 20, 17, hsi_lo=3, hsi_hi=13, device=DEV, direct_h2d=direct)
    outs = []
    for k in range(5):
        cube = _u(70 + k, (2, 12, 20, 17))
        np.copyto(st.host_slot(), cube.numpy())
        x = st.submit()
        outs.append((x.clone(), cube[..., 3:13].permute(0, 3, 1, 2).unsqueeze(1)))
        if k % 2 == 0:
            st.release()                       # odd submits exercise the conservative wait instead
    torch.cuda.synchronize()
    for got, want in outs:
        assert torch.equal(got.cpu(), want)


def test_stager_fp16_source():
    from hyperpri_amd.ingest import CubeStager
    st = CubeStager(1, 8, 10, 9, device=DEV, unsqueeze_hsi=False, src_dtype=np.float16)
    cube = _u(80, (1, 8, 10, 9)).half()
    np.copyto(st.host_slot(), cube.numpy())
    x = st.submit()
    assert x.shape == (1, 9, 8, 10) and torch.equal(x.cpu(), cube.float().permute(0, 3, 1, 2))


def test_cubenet_consumes_staged_cube_in_place_bit_identical():
    import hyperpri_amd as H
    from hyperpri_amd.ingest import CubeStager
    net = H.CubeNET(6, 1, first_depth=64, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = net.to(DEV).train()
    hwb = _u(1235, (2, 36, 50, 9))                                       # 9 bands on "disk", the net takes [2:8]
    mask = (_u(4321, (2, 1, 36, 50)) > 0.9).float().to(DEV)
    x_ref = hwb[..., 2:8].permute(0, 3, 1, 2).unsqueeze(1).contiguous().to(DEV)    # what dataset.py would hand over
    loss = torch.nn.BCEWithLogitsLoss()(net(x_ref), mask)
    loss.backward()
    want_logits = net(x_ref).detach().clone()
    want_grads = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    st = CubeStager(2, 36, 50, 9, hsi_lo=2, hsi_hi=8, device=DEV)
    np.copyto(st.host_slot(), hwb.numpy())
    x = st.submit()
    assert x.shape == x_ref.shape and torch.equal(x, x_ref)
    import hyperpri_amd.engine as E
    calls = []
    orig = E._lib.call
    E._lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        loss2 = torch.nn.BCEWithLogitsLoss()(net(x), mask)
    finally:
        E._lib.call = orig
    assert "hpri_nchw_to_nhwc" not in calls                                # consumed in place
    loss2.backward()
    st.release()
    assert torch.equal(loss2.detach(), loss.detach())
    for g, w in zip((p.grad for p in net.parameters()), want_grads):
        assert torch.equal(g, w)
    assert torch.equal(net(x).detach(), want_logits)


def test_spectral_unet_takes_4d_staged_cube():
    import hyperpri_amd as H
    from hyperpri_amd.ingest import from_hwb
    net = H.SpectralUNET(22, 1, 48)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = net.to(DEV).eval()
    hwb = _u(1238, (2, 12, 20, 22))
    with torch.no_grad():
        a = net(hwb.permute(0, 3, 1, 2).contiguous().to(DEV))
        b = net(from_hwb(hwb.to(DEV), unsqueeze_hsi=False))
    assert torch.equal(a, b)
