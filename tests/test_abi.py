"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol the header
declares; argument validation returns error codes instead of launching (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_every_declared_symbol():
    from hyperpri_amd import _lib
    decls = _lib.parse_header()
    names = set(re.findall(r"\b(hpri_\w+)\s*\(", open(os.path.join(ROOT, "include", "hyperpri_hip.h")).read()))
    assert names and names == set(decls.keys())
    lib = _lib.load()
    for n in names:
        assert hasattr(lib, n), n
    assert lib.hpri_version() >= 100


def test_bad_arguments_are_rejected_without_launch():
    from hyperpri_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    rc = lib.hpri_conv_fwd(null, 8, 0, null, null, null, 8, 0, null, 1, 4, 4, 8, 8, 64, 8, 3, 0, 0, 0, 0, 0, 0, 0, 0, null, 0, null)
    assert rc == -1 and b"null" in lib.hpri_last_error()
    rc = lib.hpri_maxpool2_fwd(null, 8, 0, null, 8, 0, 1, 4, 4, 8, null)
    assert rc == -1
    with pytest.raises(RuntimeError):
        _lib.call("hpri_fill", null, 0, 0.0, null)


def test_plans_are_pure_host_functions():
    from hyperpri_amd import _lib
    lib = _lib.load()
    s, cr, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.hpri_wgrad_plan(2, 608, 968, 64, 64, 3, ctypes.byref(s), ctypes.byref(cr), ctypes.byref(nr)) == 0
    assert s.value >= 1 and cr.value == 64 and nr.value == 64
    k, t, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    assert lib.hpri_conv_fwd_plan(2, 608, 968, 64, 64, 3, 0, 0, ctypes.byref(k), ctypes.byref(t), ctypes.byref(w)) == 0
    assert (k.value, t.value, w.value) == (1, 2 * (76 * 30 + 19), 0)   # 30 columns of 8x32 tiles + one of 32x8
    assert lib.hpri_conv_fwd_plan(2, 38, 60, 1024, 1024, 3, 0, 0, ctypes.byref(k), ctypes.byref(t), ctypes.byref(w)) == 0
    assert k.value > 1 and t.value == 2 * 36 and w.value == k.value * 2 * 38 * 60 * 1024
    assert lib.hpri_packed_weight_floats(238, 64, 9) == 8 * 9 * 32 * 64


def test_modules_keep_reference_state_dict_and_init():
    import json
    import torch
    import hyperpri_amd as H
    known = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))
    for nm, m in [("unet3_full", H.UNet(3, 1, bilinear=False)), ("cubenet64_full", H.CubeNET(238, 1, 64, bilinear=False)),
                  ("cubenet128_full", H.CubeNET(300, 1, 128, bilinear=False)),
                  ("spectral1650_full", H.SpectralUNET(238, 1, 1650))]:
        sd = m.state_dict()
        assert list(sd.keys()) == known[nm]["keys"]
        assert [list(v.shape) for v in sd.values()] == known[nm]["shapes"]
    c = known["counts"]
    assert sum(p.numel() for p in H.UNet(3, 1, bilinear=False).parameters()) == c["UNet(3,1)"]["elements"]
    for seed_key, mk in [("init_seed7_unet3", lambda: H.UNet(3, 1, bilinear=False)),
                         ("init_seed7_cubenet64_d6", lambda: H.CubeNET(6, 1, 64, bilinear=False)),
                         ("init_seed7_spectral_10_4", lambda: H.SpectralUNET(10, 1, 4))]:
        torch.manual_seed(7)
        sd = mk().state_dict()
        for k, v in known[seed_key].items():
            assert [float(t) for t in sd[k].flatten()[:4]] == v, (seed_key, k)
    m = H.CubeNET(6, 1, 64, bilinear=False)
    assert m.first_conv.weight is m.inc[0].weight      # aliased parameter, models.py:169-171
    assert H.initialize_model('UNET', 1, {'channels': 3, 'bilinear': False, 'feature_extraction': False,
                                          'use_attention': False}).n_channels == 3
    assert H.translate_load_dir('CubeNET', {'3d_featmaps': 64}) == 'CubeNET_64'


def test_cpu_inputs_fail_loudly():
    import torch
    import hyperpri_amd as H
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        H.OutConv(4, 1)(torch.zeros(1, 4, 2, 2))


def test_split_k_is_priced_by_its_slab_traffic():
    """Host-side planning (no GPU): the bf16 plane convolution takes K slices only below half a round of work items (every
    CubeNET-64 layer from 38x60 up runs unsliced), and the direct kernels stop slicing where the slabs outweigh whole rounds."""
    import ctypes
    from hyperpri_amd import _lib
    lib = _lib.load()
    k = ctypes.c_int(); tl = ctypes.c_int(); ws = ctypes.c_size_t()
    for (N, H, W, cin, cout) in [(2, 38, 60, 512, 1024), (2, 76, 121, 1024, 512), (2, 152, 242, 256, 256), (2, 608, 968, 256, 64)]:
        assert lib.hpri_conv_bf16v3_plan(N, H, W, cin, cout, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(ws)) == 0
        assert k.value == 1 and ws.value == 0, (H, W, cin, cout, k.value)
    assert lib.hpri_conv_bf16v3_plan(1, 16, 24, 1024, 64, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(ws)) == 0
    assert k.value > 1 and ws.value == k.value * 16 * 24 * 64       # a tiny problem: slices fill the chip


def test_plane_conv_v3_plan_matches_a_python_restatement_of_its_tiling():
    """hpri_conv_bf16v3_plan is host-only (no GPU needed): its statistics-tile count must equal the tiling the kernel documents --
    column bands of 32 x 8 and 16 x 16 pixel tiles (256 pixels either way) chosen by padding cost, two workgroup slots per CU,
    K slices only below half a round of slots, bf16 output unavailable for split problems."""
    import ctypes
    from hyperpri_amd import _lib
    lib = _lib.load()

    def cdiv(a, b):
        return (a + b - 1) // b

    def tiles_img(H, W):
        best = None
        for n32 in range(0, cdiv(W, 32) + 1):
            rem = W - n32 * 32
            if rem <= 0 and n32 * 32 - W >= 32:
                continue
            n16, n8 = (cdiv(rem, 16) if rem > 0 else 0), 0
            if n16 > 0 and rem - (n16 - 1) * 16 <= 8:        # (round 5: at most 8 columns left for the last tile column: 32 x 8 tiles)
                n16, n8 = n16 - 1, 1
            cost = cdiv(H, 8) * 8 * 32 * n32 + cdiv(H, 16) * 16 * 16 * n16 + cdiv(H, 32) * 32 * 8 * n8
            t = cdiv(H, 8) * n32 + cdiv(H, 16) * n16 + cdiv(H, 32) * n8
            if best is None or cost < best[0] or (cost == best[0] and n16 + n8 == 0):
                best = (cost, t)
        return best[1]
    for (N, H, W, cin_pad, cout_pad) in [(2, 608, 968, 256, 64), (2, 304, 484, 128, 128), (2, 152, 242, 512, 256), (1, 76, 121, 1024, 512),
                                         (2, 38, 60, 1024, 1024), (1, 17, 23, 32, 64), (3, 16, 16, 32, 320), (1, 1, 1, 32, 64)]:
        k, tl, ws = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        assert lib.hpri_conv_bf16v3_plan(N, H, W, cin_pad, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(ws)) == 0
        assert 1 <= k.value <= 4
        if k.value == 1:
            assert tl.value == N * tiles_img(H, W), (N, H, W, tl.value)
            assert ws.value == 0
        else:
            assert cin_pad // 32 // k.value >= 4                       # never fewer than four 32-channel chunks per slice
            assert tl.value == N * cdiv(H * W, 64) and ws.value == k.value * N * H * W * cout_pad
    # the Winograd plan: one record per 16 x 8-pixel tile
    t = ctypes.c_int()
    assert lib.hpri_conv_wino4_plan(2, 608, 968, ctypes.byref(t)) == 0 and t.value == 2 * 76 * 61


def test_plane_gemm_and_1x1_weight_gradient_plans_match_a_python_restatement():
    """hpri_gemm_bf16v3_plan / hpri_wgrad1x1_bf16v3_plan are host-only: statistics tiles of the GEMM never straddle an image; the
    weight gradient pads its slab to 256 n x 128 c tiles, takes a multiple of 8 pixel splits (one XCD owns whole splits) for about
    three rounds of two workgroups per CU, never fewer than 64 stages of 32 pixels per split, and a single split for small problems;
    the bad-argument paths of the new entry points return errors without a launch."""
    import ctypes
    from hyperpri_amd import _lib
    lib = _lib.load()

    def cdiv(a, b):
        return (a + b - 1) // b
    for N, HW in [(1, 425600), (2, 588544), (3, 257), (1, 5), (4, 256)]:
        t = ctypes.c_int()
        assert lib.hpri_gemm_bf16v3_plan(N, HW, ctypes.byref(t)) == 0 and t.value == N * cdiv(HW, 256)
    assert lib.hpri_gemm_bf16v3_plan(0, 10, ctypes.byref(ctypes.c_int())) != 0
    slots = 2 * 256                                             # (no device here: the compute-unit count falls back to 256)
    for P, cin_pad, cout_pad in [(425600, 1664, 1664), (425600, 3328, 1664), (425600, 256, 1664), (1177088, 64, 64), (300, 64, 64), (5000, 256, 192),
                                 (2 * 304 * 484, 128, 256)]:
        sp, cr, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert lib.hpri_wgrad1x1_bf16v3_plan(P, cin_pad, cout_pad, ctypes.byref(sp), ctypes.byref(cr), ctypes.byref(nr)) == 0
        assert cr.value == cdiv(cin_pad, 128) * 128 and nr.value == cdiv(cout_pad, 256) * 256
        tiles = (cr.value // 128) * (nr.value // 256)
        stages = cdiv(P, 32)
        s = cdiv(cdiv(3 * slots, tiles), 8) * 8
        while s > 8 and stages // s < 64:
            s -= 8
        if s == 8 and stages // 8 < 16:
            s = 1
        per = cdiv(stages, s)
        assert sp.value == cdiv(stages, per), (P, cin_pad, cout_pad, sp.value, s)
        assert sp.value == 1 or sp.value % 8 == 0 or cdiv(stages, per) < s          # (rounding may drop empty trailing splits)
    null = ctypes.c_void_p(0)
    assert lib.hpri_gemm_bf16v3(null, 64, 0, null, null, null, 0, 0, null, 0, 0, null, 0, 1, 256, 64, 64, 64, 64, 0, null) != 0
    assert lib.hpri_convt_fwd_bf16v3(null, 64, 0, null, null, null, 0, 0, null, 0, 0, 1, 4, 4, 64, 16, 64, 8, 8, 0, 0, null) != 0
    assert lib.hpri_convt_dgrad_bf16v3(null, 64, 0, null, null, 64, 0, 1, 4, 4, 32, 64, 64, 64, 8, 8, 0, 0, 0, null) != 0
    assert lib.hpri_wgrad1x1_bf16v3(null, 64, 0, 64, null, 64, 0, 64, null, 0, 100, 64, 64, null) != 0
    assert lib.hpri_wgrad_convt_bf16v3(null, 64, 0, 64, null, 64, 0, null, 0, 1, 4, 4, 64, 64, 8, 8, 0, 0, null) != 0
    assert lib.hpri_conv_bf16v3_y2(null, 64, 0, null, null, null, 64, 0, null, 1, 8, 8, 64, 64, 64, 64, null, 64, 0, 0, 64, 0, null) != 0
    assert lib.hpri_pack_weight_bf16_gap(null, null, 0, 64, 64, 64, 1, 60, 30, 4, null) != 0
    assert b"null" in lib.hpri_last_error() or len(lib.hpri_last_error()) > 0
