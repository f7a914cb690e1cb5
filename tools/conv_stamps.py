#!/usr/bin/env python3
"""Diagnostic: run the stamped conv_fwd variant (experiments/lib_v8.so) on one shape and print where a wave's cycles go."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tools.conv_bench import load
lib = load(sys.argv[1])
lib.hpri_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (N, H, W, Cin, Cout, ks) in [(2, 304, 484, 128, 128, 3), (2, 608, 968, 64, 64, 3)]:
    x = torch.randn(N * H * W * Cin, device=dev); w = torch.randn(Cout * Cin * 9, device=dev) * 0.05
    b = torch.randn(Cout, device=dev); y = torch.empty(N * H * W * Cout, device=dev)
    cp = (Cout + 63) // 64 * 64
    wp = torch.empty(lib.hpri_packed_weight_floats(Cin, cp, 9), device=dev)
    lib.hpri_pack_weight(P(w), P(wp), 0, Cin, Cout, cp, 9, 0, 0, Cin, st)
    stats = torch.empty(lib.hpri_conv_fwd_tiles(N, H, W, cp) * cp * 4, device=dev)
    for _ in range(3):
        lib.hpri_conv_fwd(P(x), Cin, 0, P(wp), P(b), P(y), Cout, 0, P(stats), N, H, W, Cin, Cout, cp, Cout, ks, 0, 0, 0, 0, 0, 0, 0, 0, st)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    lib.hpri_debug_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
    d = buf.reshape(-1, 8)[:2048 * 4]
    d = d[d[:, 6] > 0].astype(np.float64)
    S = d[0, 6]
    print(f"shape {N}x{H}x{W} {Cin}->{Cout}: waves={len(d)} panels={int(S)}  per-panel cycles (mean over waves):")
    names = ["total loop", "store(+vmcnt wait)", "barrier", "load issue", "mfma phase", "epilogue"]
    for i, n in enumerate(names):
        v = d[:, i].mean() / (S if i in (1, 2, 3, 4) else 1)
        print(f"   {n:20s} {v:10.0f}" + (f"   ({d[:, i].mean() / d[:, 0].mean() * 100:5.1f} % of loop)" if i in (1, 2, 3, 4) else ""))
    print(f"   ideal mfma cycles/panel at 2 waves/SIMD sharing: {64 * 64 * 2}")
