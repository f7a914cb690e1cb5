#!/usr/bin/env python3
"""Winograd F(2x2,3x3) fp32 kernel (conv_wino.hip) against the direct fp32 MFMA kernel: outputs, BatchNorm partial
statistics (as per-channel mean / variance), accumulate + ReLU epilogues, data-gradient pack, odd geometries; and an
interleaved A/B timing on the CubeNET layer shapes.   usage: wino_check.py [out.json]"""
import os as _os
_os.environ.setdefault("HPRI_DIAG", "1")     # uses entry points of the DIAGNOSTICS build (include/hyperpri_hip_diag.h): HPRI_DIAG=1 python -m hyperpri_amd.build
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402

CHECK = [(1, 17, 23, 5, 7), (2, 36, 50, 64, 64), (1, 76, 121, 128, 192), (2, 38, 60, 40, 64), (1, 16, 16, 8, 64)]
BENCH = [(2, 608, 968, 238, 64), (2, 608, 968, 64, 64), (2, 608, 968, 128, 64), (2, 304, 484, 64, 128), (2, 304, 484, 128, 128),
         (2, 304, 484, 256, 128), (2, 152, 242, 256, 256), (2, 152, 242, 512, 256), (2, 76, 121, 512, 512), (2, 76, 121, 1024, 512), (2, 38, 60, 512, 1024), (2, 38, 60, 1024, 1024)]


def rup(x, m):
    return (x + m - 1) // m * m


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    rows = []

    def setup(N, H, W, Cin, Cout, mode):
        cs, cout_pad = rup(Cin, 8), rup(Cout, 64)
        x = torch.zeros(N * H * W, cs, device=dev)
        x[:, :Cin] = torch.randn(N * H * W, Cin, device=dev)
        if mode == 0:
            w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.1
            K, ncols, d1 = Cin, Cout, Cin
        else:                      # data gradient of a layer with Cout -> "Cin" here: weight (K=Cin_here outputs ... )
            w = torch.randn(Cin, Cout, 3, 3, device=dev) * 0.1      # W[n = k][c = col]
            K, ncols, d1 = Cin, Cout, Cout
        b = torch.randn(Cout, device=dev)
        wp = torch.empty(lib.hpri_packed_weight_floats(K, cout_pad, 9), device=dev)
        assert lib.hpri_pack_weight(P(w), P(wp), mode, K, ncols, cout_pad, 9, 0, 0, d1, st) == 0
        up = torch.empty(lib.hpri_wino_packed_floats(K, cout_pad), device=dev)
        assert lib.hpri_wino_pack(P(w), P(up), P(None), mode, K, ncols, cout_pad, d1, st) == 0
        up4 = torch.empty(lib.hpri_wino_packed_floats(K, cout_pad), device=dev)
        assert lib.hpri_wino4_pack(P(w), P(up4), P(None), mode, K, ncols, cout_pad, d1, st) == 0
        return cs, cout_pad, x, w, b, wp, (up, up4)

    def run_direct(x, cs, wp, b, y, stats, N, H, W, Cout, cout_pad, acc):
        k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        lib.hpri_conv_fwd_plan(N, H, W, cs, cout_pad, 3, 0, 0, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
        ws = torch.empty(max(wsf.value, 4), device=dev)
        sts = torch.zeros(tl.value * cout_pad * 4, device=dev) if stats else None
        rc = lib.hpri_conv_fwd(P(x), cs, 0, P(wp), P(b), P(y), rup(Cout, 8), 0, P(sts), N, H, W, cs, Cout, cout_pad, rup(Cout, 8), 3, 0, 0, acc,
                               0, 0, 0, 0, 0, P(ws), ws.numel(), st)
        assert rc == 0, lib.hpri_last_error()
        return sts, tl.value

    def run_wino(x, cs, up, b, y, stats, N, H, W, Cout, cout_pad, acc):
        tl = ctypes.c_int()
        lib.hpri_conv_wino_plan(N, H, W, ctypes.byref(tl))
        sts = torch.zeros(tl.value * cout_pad * 4, device=dev) if stats else None
        rc = lib.hpri_conv_wino(P(x), cs, 0, P(up), P(b), P(y), rup(Cout, 8), 0, P(sts), N, H, W, cs, Cout, cout_pad, rup(Cout, 8), acc, st)
        assert rc == 0, lib.hpri_last_error()
        return sts, tl.value

    def run_wino4(x, cs, up, b, y, stats, N, H, W, Cout, cout_pad, acc):
        tl = ctypes.c_int()
        lib.hpri_conv_wino4_plan(N, H, W, ctypes.byref(tl))
        sts = torch.zeros(tl.value * cout_pad * 4, device=dev) if stats else None
        rc = lib.hpri_conv_wino4(P(x), cs, 0, P(up), P(b), P(y), rup(Cout, 8), 0, P(sts), N, H, W, cs, Cout, cout_pad, rup(Cout, 8), acc, st)
        assert rc == 0, lib.hpri_last_error()
        return sts, tl.value

    def chan_stats(stats, tiles, cout_pad, Cout):
        s = stats.view(tiles, cout_pad, 4).double()
        n = s[:, :, 2]
        mean = (s[:, :, 0] * n).sum(0) / n.sum(0)
        m2 = (s[:, :, 1] + n * (s[:, :, 0] - mean) ** 2).sum(0)
        return mean[:Cout], (m2 / n.sum(0))[:Cout]

    ok = True
    for (N, H, W, Cin, Cout) in CHECK:
        for mode in (0, 1):
            cs, cout_pad, x, w, b, wp, (up, up4) = setup(N, H, W, Cin, Cout, mode)
            xt = x[:, :Cin].reshape(N, H, W, Cin).permute(0, 3, 1, 2).double()
            wt = w.double() if mode == 0 else w.double().permute(1, 0, 2, 3).flip(2, 3)
            ref = torch.nn.functional.conv2d(xt, wt, b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
            for acc in (0, 2, 1):
                y0 = torch.full((N * H * W * rup(Cout, 8),), 0.25, device=dev)
                y1 = y0.clone()
                s0, t0 = run_direct(x, cs, wp, b, y0, acc != 1, N, H, W, Cout, cout_pad, acc)
                s1, t1 = run_wino(x, cs, up, b, y1, acc != 1, N, H, W, Cout, cout_pad, acc)
                y4 = torch.full_like(y0, 0.25)
                s4, t4 = run_wino4(x, cs, up4, b, y4, acc != 1, N, H, W, Cout, cout_pad, acc)
                want = ref.clamp(min=0) if acc == 2 else (ref + 0.25 if acc == 1 else ref)
                e0 = float((y0.view(-1, rup(Cout, 8))[:, :Cout].double() - want).abs().max())
                e1 = float((y1.view(-1, rup(Cout, 8))[:, :Cout].double() - want).abs().max())
                line = f"check N{N} {H}x{W} {Cin}->{Cout} mode{mode} acc{acc}: |direct-fp64| {e0:.2e}  |wino-fp64| {e1:.2e}  scale {float(want.abs().max()):.1f}"
                if acc != 1:
                    m0, v0 = chan_stats(s0, t0, cout_pad, Cout)
                    m1, v1 = chan_stats(s1, t1, cout_pad, Cout)
                    dv = float(((v0 - v1).abs() / (v0 + 1e-3)).max())     # (ReLU outputs can have ~zero variance)
                    line += f"  dmean {float((m0 - m1).abs().max()):.1e} dvar_rel {dv:.1e}"
                    good = float((m0 - m1).abs().max()) < 1e-4 and dv < 1e-4
                    ok &= good
                    if not good: line += "   <-- STATISTICS DIFFER"
                good = e1 < 5e-5 * max(1.0, float(want.abs().max()))
                ok &= good
                if not good: line += "   <-- OUTPUT DIFFERS"
                e4 = float((y4.view(-1, rup(Cout, 8))[:, :Cout].double() - want).abs().max())
                line += f"  |wino4-fp64| {e4:.2e}"
                good = e4 < 5e-5 * max(1.0, float(want.abs().max()))
                if acc != 1:
                    m4, v4 = chan_stats(s4, t4, cout_pad, Cout)
                    good &= float((m0 - m4).abs().max()) < 1e-4 and float(((v0 - v4).abs() / (v0 + 1e-3)).max()) < 1e-4
                ok &= good
                if not good: line += "   <-- WINO4 DIFFERS"
                print(line, flush=True)
    # ---- weight gradient ----
    for (N, H, W, Cin, Cout) in CHECK + [(2, 76, 121, 64, 128)]:
        cs, cout_pad = rup(Cin, 8), rup(Cout, 64)
        x = torch.zeros(N * H * W, cs, device=dev); x[:, :Cin] = torch.randn(N * H * W, Cin, device=dev)
        cso = rup(Cout, 8)
        dy = torch.zeros(N * H * W, cso, device=dev); dy[:, :Cout] = torch.randn(N * H * W, Cout, device=dev)
        xt = x[:, :Cin].reshape(N, H, W, Cin).permute(0, 3, 1, 2).double()
        dt = dy[:, :Cout].reshape(N, H, W, Cout).permute(0, 3, 1, 2).double()
        ref = torch.nn.grad.conv2d_weight(xt, (Cout, Cin, 3, 3), dt, padding=1)
        sp, cr, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.hpri_wino_wgrad_plan(N, H, W, cs, cout_pad, ctypes.byref(sp), ctypes.byref(cr), ctypes.byref(nr))
        ws = torch.empty(sp.value * 16 * cr.value * nr.value, device=dev)
        dw = torch.full((Cout, Cin, 3, 3), 0.5, device=dev)
        for acc in (0, 1):
            rc = lib.hpri_conv_wino_wgrad(P(x), cs, 0, cs, P(dy), cso, 0, cso, P(ws), ws.numel(), N, H, W, cs, cout_pad, st)
            assert rc == 0, lib.hpri_last_error()
            assert lib.hpri_wino_wgrad_reduce(P(ws), P(dw), N, H, W, Cin, cs, Cout, cout_pad, acc, st) == 0
        e = float((dw.double() - 2 * ref).abs().max())       # second call accumulated onto the first
        sc = float(ref.abs().max())
        print(f"check wgrad N{N} {H}x{W} {Cin}->{Cout}: splits {sp.value}  |wino - fp64| {e:.2e} of {sc:.1f}", flush=True)
        ok &= e < 2e-5 * max(1.0, sc) * 2
    print("CHECK", "PASSED" if ok else "FAILED", flush=True)
    for (N, H, W, Cin, Cout) in BENCH:
        cs, cout_pad, x, w, b, wp, (up, up4) = setup(N, H, W, Cin, Cout, 0)
        y = torch.empty(N * H * W * rup(Cout, 8), device=dev)
        flops = 2.0 * N * H * W * Cin * Cout * 9
        res = {"direct": [], "wino": [], "wino4": []}
        for rnd in range(5):
            for kind in ("direct", "wino", "wino4"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    {"direct": run_direct, "wino": run_wino, "wino4": run_wino4}[kind](x, cs, {"direct": wp, "wino": up, "wino4": up4}[kind], b, y, True, N, H, W, Cout, cout_pad, 0)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    res[kind].append(e0.elapsed_time(e1) / 4)
        # weight gradient A/B
        dyb = torch.randn(N * H * W, Cout, device=dev)
        s_, cr_, nr_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.hpri_wgrad_plan(N, H, W, cs, cout_pad, 3, ctypes.byref(s_), ctypes.byref(cr_), ctypes.byref(nr_))
        wsd = torch.empty(s_.value * 9 * cr_.value * nr_.value, device=dev)
        lib.hpri_wino_wgrad_plan(N, H, W, cs, cout_pad, ctypes.byref(s_), ctypes.byref(cr_), ctypes.byref(nr_))
        wsw = torch.empty(s_.value * 16 * cr_.value * nr_.value, device=dev)
        dwt = torch.empty(Cout, Cin, 3, 3, device=dev)
        resw = {"direct": [], "wino": []}
        for rnd in range(4):
            for kind in ("direct", "wino"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    if kind == "direct":
                        assert lib.hpri_conv_wgrad(P(x), cs, 0, cs, P(dyb), Cout, 0, Cout, P(wsd), wsd.numel(), N, H, W, cs, cout_pad, 3, 0, 0, 0, 0, 0, 0, st) == 0
                        lib.hpri_wgrad_reduce(P(wsd), P(dwt), N, H, W, Cin, cs, Cout, cout_pad, 3, 0, 0, 0, st)
                    else:
                        assert lib.hpri_conv_wino_wgrad(P(x), cs, 0, cs, P(dyb), Cout, 0, Cout, P(wsw), wsw.numel(), N, H, W, cs, cout_pad, st) == 0
                        lib.hpri_wino_wgrad_reduce(P(wsw), P(dwt), N, H, W, Cin, cs, Cout, cout_pad, 0, st)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    resw[kind].append(e0.elapsed_time(e1) / 3)
        mw = {k: sorted(v)[len(v) // 2] for k, v in resw.items()}
        print(f"   wgrad (+reduce): direct {mw['direct']:.3f} ms ({flops / mw['direct'] / 1e9:.1f} TF)   winograd {mw['wino']:.3f} ms "
              f"({flops / mw['wino'] / 1e9:.1f} effective TF)  x{mw['direct'] / mw['wino']:.2f}  splits {s_.value}", flush=True)
        del dyb, wsd, wsw
        md = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        rows.append({"shape": [N, H, W, Cin, Cout], "direct_ms": md["direct"], "wino_ms": md["wino"], "wino4_ms": md["wino4"],
                     "direct_tf": flops / md["direct"] / 1e9, "wino_effective_tf": flops / md["wino"] / 1e9,
                     "wino4_effective_tf": flops / md["wino4"] / 1e9})
        print(f"N{N} {H}x{W} {Cin}->{Cout}: direct {md['direct']:.3f} ms ({rows[-1]['direct_tf']:.1f} TF)   winograd {md['wino']:.3f} ms "
              f"({rows[-1]['wino_effective_tf']:.1f} effective TF)   wino4 {md['wino4']:.3f} ms ({rows[-1]['wino4_effective_tf']:.1f})  "
              f"x{md['wino'] / md['wino4']:.3f} vs wino", flush=True)
        del x, y
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        json.dump({"check_passed": ok, "bench": rows}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
