#!/usr/bin/env python3
"""Micro-benchmark of the conv_fwd / conv_wgrad launchers for A/B-ing kernel variants: each argument is a
libhyperpri_hip.so build; all are timed interleaved in one process on the same random data (rule 24)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402

import json
SPLIT = int(os.environ.get("SPLIT", "0"))
SHAPES = json.loads(os.environ["SHAPES"]) if "SHAPES" in os.environ else [  # N, H, W, Cin, Cout, ks
    (2, 304, 484, 128, 128, 3),
    (2, 608, 968, 64, 64, 3),
    (2, 76, 121, 512, 512, 3),
    (2, 38, 60, 1024, 1024, 3),
]


def load(path):
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _lib.parse_header().items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def main():
    libs = [(os.path.basename(p), load(p)) for p in sys.argv[1:]]
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    mode = os.environ.get("MODE", "fwd")
    for (N, H, W, Cin, Cout, ks) in SHAPES:
        x = torch.randn(N * H * W * Cin, device=dev)
        w = torch.randn(Cout * Cin * ks * ks, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        y = torch.empty(N * H * W * Cout, device=dev)
        cout_pad = (Cout + 63) // 64 * 64
        flops = 2.0 * N * H * W * Cin * Cout * ks * ks
        res = {}
        preps = []
        for name, lib in libs:
            nfl = lib.hpri_packed_weight_floats(Cin, cout_pad, ks * ks)
            wp = torch.empty(nfl, device=dev)
            assert lib.hpri_pack_weight(P(w), P(wp), 0, Cin, Cout, cout_pad, ks * ks, 0, 0, Cin, st) == 0
            stats = torch.empty((N * ((H * W + 15) // 16) + 64) * cout_pad * 4, device=dev)   # >= any plan's stat_tiles
            s, cr, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            lib.hpri_wgrad_plan(N, H, W, Cin, cout_pad, ks, ctypes.byref(s), ctypes.byref(cr), ctypes.byref(nr))
            ws = torch.empty(max(s.value * ks * ks * cr.value * nr.value, 4 * N * H * W * cout_pad), device=dev)
            wpb = torch.empty(((Cin + 31) // 32) * ks * ks * cout_pad * 32 * 2, dtype=torch.bfloat16, device=dev)
            if hasattr(lib, "hpri_pack_weight_bf16"):
                assert lib.hpri_pack_weight_bf16(P(w), P(wpb), 0, Cin, Cout, cout_pad, ks * ks, Cin, 0, SPLIT, st) == 0
            preps.append((wp, stats, ws, wpb))
        for rnd in range(5):
            for (name, lib), (wp, stats, ws, wpb) in zip(libs, preps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 5
                e0.record()
                for _ in range(reps):
                    if mode == "fwd":
                        rc = lib.hpri_conv_fwd(P(x), Cin, 0, P(wp), P(b), P(y), Cout, 0, P(stats), N, H, W, Cin, Cout, cout_pad,
                                               Cout, ks, 0, 0, 0, 0, 0, 0, 0, 0, P(ws), ws.numel(), st)
                    elif mode == "fwd_bf16":
                        rc = lib.hpri_conv_fwd_bf16(P(x), Cin, 0, P(wpb), P(b), P(y), Cout, 0, P(stats), N, H, W, Cin, Cout,
                                                    cout_pad, Cout, ks, 0, 0, 0, 0, 0, 0, 0, 0, SPLIT, P(ws), ws.numel(), st)
                    elif mode == "wgrad_bf16":
                        rc = lib.hpri_conv_wgrad_bf16(P(x), Cin, 0, Cin, P(y), Cout, 0, Cout, P(ws), ws.numel(), N, H, W, Cin,
                                                      cout_pad, ks, 0, 0, 0, 0, 0, 0, SPLIT, st)
                    else:
                        rc = lib.hpri_conv_wgrad(P(x), Cin, 0, Cin, P(y), Cout, 0, Cout, P(ws), ws.numel(), N, H, W, Cin, cout_pad,
                                                 ks, 0, 0, 0, 0, 0, 0, st)
                    assert rc == 0, lib.hpri_last_error()
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    res.setdefault(name, []).append(e0.elapsed_time(e1) / reps)
        if os.environ.get("CHECK") and mode in ("fwd", "fwd_bf16"):
            outs = []
            for (name, lib), (wp, stats, ws, wpb) in zip(libs, preps):
                yy = torch.zeros_like(y)
                if mode == "fwd":
                    lib.hpri_conv_fwd(P(x), Cin, 0, P(wp), P(b), P(yy), Cout, 0, P(stats), N, H, W, Cin, Cout, cout_pad,
                                      Cout, ks, 0, 0, 0, 0, 0, 0, 0, 0, P(ws), ws.numel(), st)
                else:
                    lib.hpri_conv_fwd_bf16(P(x), Cin, 0, P(wpb), P(b), P(yy), Cout, 0, P(stats), N, H, W, Cin, Cout, cout_pad,
                                           Cout, ks, 0, 0, 0, 0, 0, 0, 0, 0, SPLIT, P(ws), ws.numel(), st)
                torch.cuda.synchronize()
                outs.append((yy, stats.clone()))
            for (name, _), (yy, stt) in zip(libs[1:], outs[1:]):
                print(f"   check {name}: max|dy| = {float((yy - outs[0][0]).abs().max()):.3e}, max|dstats| = "
                      f"{float((stt - outs[0][1]).abs().max()):.3e}")
        line = f"{mode} N{N} {H}x{W} {Cin}->{Cout} k{ks}: "
        for name, _ in libs:
            ms = sorted(res[name])
            line += f" {name}: med {flops / ms[len(ms) // 2] / 1e9:6.1f} TF (min-time {flops / ms[0] / 1e9:6.1f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
