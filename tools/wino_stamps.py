#!/usr/bin/env python3
"""Cycle shares of a conv_wino workgroup (diagnostic build, tools/build_stamps.sh): main loop | output transform + stores |
statistics, median over workgroups."""
import os as _os
_os.environ.setdefault("HPRI_DIAG", "1")     # uses entry points of the DIAGNOSTICS build (include/hyperpri_hip_diag.h): HPRI_DIAG=1 python -m hyperpri_amd.build
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
SHAPES = [(2, 608, 968, 238, 64), (2, 304, 484, 128, 128)]
def rup(x, m): return (x + m - 1) // m * m
LIBNAME = os.environ.get("WINO_LIB", "libv2stamps.so")
print("library", LIBNAME)
lib = ctypes.CDLL(os.path.join(ROOT, "hyperpri_amd", "lib", LIBNAME))
lib.hpri_last_error.restype = ctypes.c_char_p
lib.hpri_wino_packed_floats.restype = ctypes.c_size_t
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (N, H, W, Cin, Cout) in SHAPES:
    cs, cout_pad = rup(Cin, 8), rup(Cout, 64)
    x = torch.zeros(N * H * W, cs, device=dev); x[:, :Cin] = torch.randn(N * H * W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.1
    b = torch.randn(Cout, device=dev)
    up = torch.empty(lib.hpri_wino_packed_floats(Cin, cout_pad), device=dev)
    assert lib.hpri_wino_pack(P(w), P(up), ctypes.c_void_p(0), 0, Cin, Cout, cout_pad, Cin, st) == 0
    tl = ctypes.c_int(); lib.hpri_conv_wino_plan(N, H, W, ctypes.byref(tl))
    stats = torch.zeros(tl.value * cout_pad * 4, device=dev)
    y = torch.zeros(N * H * W * Cout, device=dev)
    nwg = tl.value * (cout_pad // 64)
    stamps = torch.zeros(nwg * 2 * 8, dtype=torch.int64, device=dev)
    lib.hpri_wino_set_stamps(P(stamps))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        if it == 2:
            e0.record()
        rc = lib.hpri_conv_wino(P(x), cs, 0, P(up), P(b), P(y), Cout, 0, P(stats), N, H, W, cs, Cout, cout_pad, Cout, 0, st)
        assert rc == 0, lib.hpri_last_error()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = stamps.view(-1, 2, 8).cpu().double()
    names = ["main loop", "transform+store", "statistics"]
    print(f"N{N} {H}x{W} {Cin}->{Cout}: {nwg} workgroups, {Cin // 8 if Cin % 8 == 0 else cs // 8} stages")
    span = float(t[:, :, 3].max() - t[:, :, 0].min())
    print(f"   launch {ms:.3f} ms; first stamp -> last stamp {span:.0f} ticks = {span / ms / 1e6:.3f} GHz if s_memtime counts shader clocks")
    for g in (0, 1):
        d = [t[:, g, i + 1] - t[:, g, i] for i in range(3)]
        tot = t[:, g, 3] - t[:, g, 0]
        e = [t[:, g, 4] - t[:, g, 1], t[:, g, 5] - t[:, g, 4], t[:, g, 6] - t[:, g, 5], t[:, g, 7] - t[:, g, 6], t[:, g, 2] - t[:, g, 7]]
        print(f"   waves {4*g}-{4*g+3}: epilogue pass 0: exchange writes {e[0].median():.0f} | barrier {e[1].median():.0f} | reads + A^T M A + stores {e[2].median():.0f} | barrier {e[3].median():.0f} | pass 1 {e[4].median():.0f}")
        print(f"   waves {4*g}-{4*g+3}: total {tot.median():8.0f} | " + " | ".join(f"{n} {v.median():7.0f} ({100 * v.median() / tot.median():4.1f} %)" for n, v in zip(names, d)) + f" | per stage {d[0].median() / (cs // 8):6.0f}")
