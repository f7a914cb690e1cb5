#!/usr/bin/env python3
"""DMA-fed fp32 pixel GEMM (gemm1x1.hip) on the ConvTranspose2d(k=2, s=2) forms: forward (2x2 scatter) and data gradient
(2x2 gather) against torch in fp64 on the CPU, and an interleaved A/B timing against the round-1 implicit-GEMM forms on the
four Up blocks of CubeNET-64.   usage: gemm1x1_check.py [out.json]"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402

A_DIRECT, A_S2D, E_DIRECT, E_D2S = 0, 1, 0, 1
CHECK = [(1, 5, 7, 64, 32, 0, 0), (2, 9, 6, 128, 64, 1, 1), (1, 19, 30, 256, 128, 0, 1), (1, 3, 3, 40, 32, 2, 3)]   # N, h, w, Cin, Cup, dY, dX
BENCH = [(2, 38, 60, 1024, 512, 0, 1), (2, 76, 121, 512, 256, 0, 0), (2, 152, 242, 256, 128, 0, 0), (2, 304, 484, 128, 64, 0, 0)]


def rup(x, m):
    return (x + m - 1) // m * m


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())

    def setup(N, h, w, Cin, Cup, dY, dX):
        H2, W2 = 2 * h + dY, 2 * w + dX
        py0, px0 = dY // 2, dX // 2
        cs = rup(Cin, 8)
        x = torch.zeros(N * h * w, cs, device=dev); x[:, :Cin] = torch.randn(N * h * w, Cin, device=dev)
        wt = torch.randn(Cin, Cup, 2, 2, device=dev) * 0.1
        b = torch.randn(Cup, device=dev)
        ycs = rup(Cup, 8) + 16                      # a slice of a wider (concat) buffer
        return H2, W2, py0, px0, cs, x, wt, b, ycs

    def fwd_new(x, cs, wt, b, y, ycs, yoff, N, h, w, Cin, Cup, H2, W2, py0, px0):
        wp = torch.empty(lib.hpri_gemm1x1_packed_floats(Cin, 4 * Cup), device=dev)
        assert lib.hpri_gemm1x1_pack(P(wt), P(wp), 0, Cin, 4 * Cup, Cin, Cup, st) == 0
        def call():
            return lib.hpri_gemm1x1(P(x), cs, 0, x.numel(), P(wp), P(b), P(y), ycs, yoff, N, h, w, cs, 4 * Cup, Cup, A_DIRECT, E_D2S,
                                    H2, W2, py0, px0, Cup, 0, st)
        return call, wp

    def dgrad_new(g, gcs, goff, wt, gx, N, h, w, Cin, Cup, H2, W2, py0, px0, acc):
        wp = torch.empty(lib.hpri_gemm1x1_packed_floats(4 * Cup, Cin), device=dev)
        assert lib.hpri_gemm1x1_pack(P(wt), P(wp), 1, 4 * Cup, Cin, Cin, Cup, st) == 0
        gxcs = gx.shape[1]
        def call():
            return lib.hpri_gemm1x1(P(g), gcs, goff, g.numel(), P(wp), P(None), P(gx), gxcs, 0, N, h, w, 4 * Cup, Cin, rup(Cin, 4), A_S2D,
                                    E_DIRECT, H2, W2, py0, px0, Cup, acc, st)
        return call, wp

    ok = True
    for (N, h, w, Cin, Cup, dY, dX) in CHECK:
        H2, W2, py0, px0, cs, x, wt, b, ycs = setup(N, h, w, Cin, Cup, dY, dX)
        yoff = 8
        y = torch.full((N * H2 * W2, ycs), 7.0, device=dev)
        call, _wp = fwd_new(x, cs, wt, b, y, ycs, yoff, N, h, w, Cin, Cup, H2, W2, py0, px0)
        assert call() == 0, lib.hpri_last_error()
        torch.cuda.synchronize()
        xt = x[:, :Cin].reshape(N, h, w, Cin).permute(0, 3, 1, 2).double().cpu()
        ref = torch.nn.functional.conv_transpose2d(xt, wt.double().cpu(), b.double().cpu(), stride=2)      # N, Cup, 2h, 2w
        got = y.view(N, H2, W2, ycs)[:, py0:py0 + 2 * h, px0:px0 + 2 * w, yoff:yoff + Cup].permute(0, 3, 1, 2).double().cpu()
        e = float((got - ref).abs().max())
        untouched = bool((y.view(N, H2, W2, ycs)[..., :yoff] == 7.0).all() and (y.view(N, H2, W2, ycs)[..., yoff + Cup:] == 7.0).all())
        # data gradient: dX1 = conv2d-style gather of g with the same weight
        g = torch.randn(N * H2 * W2, ycs, device=dev)
        gx = torch.full((N * h * w, rup(Cin, 8)), 0.5, device=dev)
        call, _wp2 = dgrad_new(g, ycs, yoff, wt, gx, N, h, w, Cin, Cup, H2, W2, py0, px0, 1)
        assert call() == 0, lib.hpri_last_error()
        torch.cuda.synchronize()
        gt = g.view(N, H2, W2, ycs)[:, py0:py0 + 2 * h, px0:px0 + 2 * w, yoff:yoff + Cup].permute(0, 3, 1, 2).double().cpu()
        refd = torch.nn.functional.conv2d(gt, wt.double().cpu(), None, stride=2)                           # N, Cin, h, w
        gotd = gx[:, :Cin].reshape(N, h, w, Cin).permute(0, 3, 1, 2).double().cpu() - 0.5
        ed = float((gotd - refd).abs().max())
        good = e < 2e-5 * max(1.0, float(ref.abs().max())) and ed < 2e-5 * max(1.0, float(refd.abs().max())) and untouched
        ok &= good
        print(f"check N{N} {h}x{w} {Cin}->{Cup} pad({dY},{dX}): fwd |err| {e:.2e} of {float(ref.abs().max()):.1f}  dgrad |err| {ed:.2e} of "
              f"{float(refd.abs().max()):.1f}  neighbours untouched {untouched}" + ("" if good else "   <-- DIFFERS"), flush=True)
    print("CHECK", "PASSED" if ok else "FAILED", flush=True)
    if not ok:
        sys.exit(1)
    rows = []
    for (N, h, w, Cin, Cup, dY, dX) in BENCH:
        H2, W2, py0, px0, cs, x, wt, b, ycs = setup(N, h, w, Cin, Cup, dY, dX)
        y = torch.zeros(N * H2 * W2, ycs, device=dev)
        g = torch.randn(N * H2 * W2, ycs, device=dev)
        gx = torch.zeros(N * h * w, cs, device=dev)
        flops = 2.0 * N * h * w * Cin * 4 * Cup
        f_new, _k1 = fwd_new(x, cs, wt, b, y, ycs, 0, N, h, w, Cin, Cup, H2, W2, py0, px0)
        d_new, _k2 = dgrad_new(g, ycs, 0, wt, gx, N, h, w, Cin, Cup, H2, W2, py0, px0, 0)
        # round-1 forms
        ncp = rup(4 * Cup, 64)
        wpo = torch.empty(lib.hpri_packed_weight_floats(Cin, ncp, 1), device=dev)
        assert lib.hpri_pack_weight(P(wt), P(wpo), 2, Cin, 4 * Cup, ncp, 1, Cup, 0, Cup, st) == 0
        k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        lib.hpri_conv_fwd_plan(N, h, w, cs, ncp, 1, A_DIRECT, E_D2S, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
        ws = torch.empty(max(wsf.value, 4), device=dev)
        def f_old():
            return lib.hpri_conv_fwd(P(x), cs, 0, P(wpo), P(b), P(y), ycs, 0, P(None), N, h, w, cs, 4 * Cup, ncp, 4 * Cup, 1, A_DIRECT, E_D2S, 0,
                                     H2, W2, py0, px0, Cup, P(ws), ws.numel(), st)
        cpd = rup(Cin, 64)
        wpd = torch.empty(lib.hpri_packed_weight_floats(4 * Cup, cpd, 1), device=dev)
        assert lib.hpri_pack_weight(P(wt), P(wpd), 3, 4 * Cup, Cin, cpd, 1, Cup, 0, Cup, st) == 0
        lib.hpri_conv_fwd_plan(N, h, w, 4 * Cup, cpd, 1, A_S2D, E_DIRECT, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
        ws2 = torch.empty(max(wsf.value, 4), device=dev)
        def d_old():
            return lib.hpri_conv_fwd(P(g), ycs, 0, P(wpd), P(None), P(gx), cs, 0, P(None), N, h, w, 4 * Cup, Cin, cpd, cs, 1, A_S2D, E_DIRECT, 0,
                                     H2, W2, py0, px0, Cup, P(ws2), ws2.numel(), st)
        res = {"fwd_old": [], "fwd_new": [], "dgrad_old": [], "dgrad_new": []}
        fns = {"fwd_old": f_old, "fwd_new": f_new, "dgrad_old": d_old, "dgrad_new": d_new}
        for rnd in range(5):
            for kind, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    rc = fn()
                    assert rc == 0, (kind, lib.hpri_last_error())
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    res[kind].append(e0.elapsed_time(e1) / 4)
        md = {k_: sorted(v)[len(v) // 2] for k_, v in res.items()}
        rows.append({"shape": [N, h, w, Cin, Cup], **{k_ + "_ms": v for k_, v in md.items()}, **{k_ + "_tf": flops / v / 1e9 for k_, v in md.items()}})
        print(f"N{N} {h}x{w} {Cin}->4x{Cup}: fwd old {md['fwd_old']:.3f} ms ({flops / md['fwd_old'] / 1e9:.0f} TF) new {md['fwd_new']:.3f} ms "
              f"({flops / md['fwd_new'] / 1e9:.0f} TF) | dgrad old {md['dgrad_old']:.3f} ms ({flops / md['dgrad_old'] / 1e9:.0f} TF) new "
              f"{md['dgrad_new']:.3f} ms ({flops / md['dgrad_new'] / 1e9:.0f} TF)", flush=True)
    if len(sys.argv) > 1:
        json.dump({"check_passed": ok, "bench": rows}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
