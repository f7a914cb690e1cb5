#!/usr/bin/env python3
"""bf16-plane weight gradient (conv_wgrad_bf16v2.hip) against fp64 conv2d_weight of the same bf16-rounded operands, and an
interleaved A/B timing against the round-1 bf16 weight-gradient kernel on the CubeNET layer shapes.
usage: wgv2_check.py [out.json]"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402

CHECK = [(1, 17, 23, 5, 7), (2, 36, 50, 64, 64), (1, 76, 121, 128, 192), (2, 38, 60, 40, 64), (1, 4, 32, 64, 64), (1, 3, 3, 8, 8),
         (1, 33, 31, 238, 64)]
BENCH = [(2, 608, 968, 238, 64), (2, 608, 968, 64, 64), (2, 608, 968, 128, 64), (2, 304, 484, 64, 128), (2, 304, 484, 128, 128),
         (2, 304, 484, 256, 128), (2, 152, 242, 256, 256), (2, 152, 242, 512, 256), (2, 76, 121, 512, 512), (2, 76, 121, 1024, 512),
         (2, 38, 60, 512, 1024), (2, 38, 60, 1024, 1024)]


def rup(x, m):
    return (x + m - 1) // m * m


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())

    def setup(N, H, W, Cin, Cout):
        cs, cso = rup(Cin, 8), rup(Cout, 8)
        x = torch.zeros(N * H * W, cs, device=dev); x[:, :Cin] = torch.randn(N * H * W, Cin, device=dev)
        dy = torch.zeros(N * H * W, cso, device=dev); dy[:, :Cout] = torch.randn(N * H * W, Cout, device=dev)
        cs16, cso16 = rup(Cin, 32), rup(Cout, 32)
        xp = torch.empty(N * H * W * cs16, dtype=torch.bfloat16, device=dev)
        dp = torch.empty(N * H * W * cso16, dtype=torch.bfloat16, device=dev)
        assert lib.hpri_to_planes(P(x), cs, 0, P(xp), 0, cs16, 0, N * H * W, Cin, cs16, 1, st) == 0
        assert lib.hpri_to_planes(P(dy), cso, 0, P(dp), 0, cso16, 0, N * H * W, Cout, cso16, 1, st) == 0
        return cs, cso, cs16, cso16, x, dy, xp, dp

    def run_v2(xp, cs16, dp, cso16, dw, N, H, W, Cin, Cout, acc):
        sp, cr, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.hpri_wgrad_bf16v2_plan(N, H, W, cs16, rup(Cout, 64), ctypes.byref(sp), ctypes.byref(cr), ctypes.byref(nr))
        ws = torch.empty(sp.value * 9 * cr.value * nr.value, device=dev)
        rc = lib.hpri_conv_wgrad_bf16v2(P(xp), cs16, 0, cs16, P(dp), cso16, 0, cso16, P(ws), ws.numel(), N, H, W, cs16, rup(Cout, 64), st)
        assert rc == 0, lib.hpri_last_error()
        rc = lib.hpri_wgrad_reduce_ex(P(ws), P(dw), sp.value, cr.value, nr.value, Cin, Cout, 3, 0, 0, acc, st)
        assert rc == 0, lib.hpri_last_error()
        return sp.value

    ok = True
    for (N, H, W, Cin, Cout) in CHECK:
        cs, cso, cs16, cso16, x, dy, xp, dp = setup(N, H, W, Cin, Cout)
        xr = x[:, :Cin].to(torch.bfloat16).double().cpu().reshape(N, H, W, Cin).permute(0, 3, 1, 2)
        dr = dy[:, :Cout].to(torch.bfloat16).double().cpu().reshape(N, H, W, Cout).permute(0, 3, 1, 2)
        ref = torch.nn.grad.conv2d_weight(xr, (Cout, Cin, 3, 3), dr, padding=1)
        dw = torch.full((Cout, Cin, 3, 3), 0.5, device=dev)
        sp = run_v2(xp, cs16, dp, cso16, dw, N, H, W, Cin, Cout, 0)
        run_v2(xp, cs16, dp, cso16, dw, N, H, W, Cin, Cout, 1)
        torch.cuda.synchronize()
        e = float((dw.double().cpu() - 2 * ref).abs().max())
        sc = max(1.0, float(ref.abs().max()))
        good = e < 4e-5 * sc
        ok &= good
        print(f"check N{N} {H}x{W} {Cin}->{Cout}: splits {sp}  |v2 - fp64(bf16 operands)| {e:.2e} of {sc:.1f}" + ("" if good else "   <-- DIFFERS"), flush=True)
    print("CHECK", "PASSED" if ok else "FAILED", flush=True)
    if not ok:
        sys.exit(1)
    rows = []
    for (N, H, W, Cin, Cout) in BENCH:
        cs, cso, cs16, cso16, x, dy, xp, dp = setup(N, H, W, Cin, Cout)
        cout_pad = rup(Cout, 64)
        flops = 2.0 * N * H * W * Cin * Cout * 9
        dwt = torch.empty(Cout, Cin, 3, 3, device=dev)
        s_, cr_, nr_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.hpri_wgrad_plan(N, H, W, cs, cout_pad, 3, ctypes.byref(s_), ctypes.byref(cr_), ctypes.byref(nr_))
        wsd = torch.empty(s_.value * 9 * cr_.value * nr_.value, device=dev)
        res = {"old": [], "v2": []}
        for rnd in range(5):
            for kind in ("old", "v2"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    if kind == "old":
                        assert lib.hpri_conv_wgrad_bf16(P(x), cs, 0, cs, P(dy), cso, 0, cso, P(wsd), wsd.numel(), N, H, W, cs, cout_pad, 3, 0, 0, 0, 0, 0, 0, 0, st) == 0
                        lib.hpri_wgrad_reduce(P(wsd), P(dwt), N, H, W, Cin, cs, Cout, cout_pad, 3, 0, 0, 0, st)
                    else:
                        sp = run_v2(xp, cs16, dp, cso16, dwt, N, H, W, Cin, Cout, 0)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    res[kind].append(e0.elapsed_time(e1) / 3)
        md = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        rows.append({"shape": [N, H, W, Cin, Cout], "old_ms": md["old"], "v2_ms": md["v2"], "old_tf": flops / md["old"] / 1e9,
                     "v2_tf": flops / md["v2"] / 1e9, "splits_v2": sp})
        print(f"N{N} {H}x{W} {Cin}->{Cout}: old {md['old']:.3f} ms ({rows[-1]['old_tf']:.0f} TF)   planes {md['v2']:.3f} ms "
              f"({rows[-1]['v2_tf']:.0f} TF)  x{md['old'] / md['v2']:.2f}  splits {sp}   (both incl. the slab reduce)", flush=True)
        del x, dy, xp, dp, wsd
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        json.dump({"check_passed": ok, "bench": rows}, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
