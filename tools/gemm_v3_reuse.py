#!/usr/bin/env python3
"""Does re-read traffic bound gemm_bf16v3 / wgrad1x1_bf16v3 on the C3 shapes?  The same K with fewer column blocks (less re-reading of
the pixel rows) and the weight gradient with fewer tiles per split.  usage: python tools/gemm_v3_reuse.py"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["ONLY"] = "none"
import torch
from hyperpri_amd import _lib
lib = _lib.load()
DEV = "cuda:0"
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rup = lambda x, m: (x + m - 1) // m * m


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


M = 608 * 700
for K, C in [(1650, 128), (1650, 512), (1650, 1650), (3300, 128), (3300, 1650)]:
    kp, cp, cw = rup(K, 32), rup(C, 64), rup(C, 4)
    xp = (torch.rand(M, kp, device=DEV) - 0.5).to(torch.bfloat16)
    wp = (torch.rand((kp // 32) * cp * 32, device=DEV) - 0.5).to(torch.bfloat16)
    y2 = torch.empty(M, cw, dtype=torch.bfloat16, device=DEV)
    ms = timeit(lambda: lib.hpri_gemm_bf16v3(P(xp), kp, 0, P(wp), P(None), P(None), 0, 0, P(y2), cw, 0, P(None), 0, 1, M, kp, C, cp, cw, 0, st()))
    print(json.dumps({"op": "gemm", "K": K, "N": C, "ms": round(ms, 3), "TF": round(2.0 * M * K * C / ms / 1e9, 1)}), flush=True)
    del xp, wp, y2
for Cin, Cout in [(128, 256), (512, 512), (1650, 1650)]:
    kx, ky = rup(Cin, 32), rup(Cout, 32)
    xp = (torch.rand(M, kx, device=DEV) - 0.5).to(torch.bfloat16)
    yp = (torch.rand(M, ky, device=DEV) - 0.5).to(torch.bfloat16)
    sp, cr, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.hpri_wgrad1x1_bf16v3_plan(M, kx, rup(Cout, 64), ctypes.byref(sp), ctypes.byref(cr), ctypes.byref(nr))
    ws = torch.empty(sp.value * cr.value * nr.value, device=DEV)
    ms = timeit(lambda: lib.hpri_wgrad1x1_bf16v3(P(xp), kx, 0, rup(Cin, 8), P(yp), ky, 0, rup(Cout, 8), P(ws), ws.numel(), M, kx, rup(Cout, 64), st()))
    print(json.dumps({"op": "wgrad", "Cin": Cin, "Cout": Cout, "splits": sp.value, "ms": round(ms, 3), "TF": round(2.0 * M * Cin * Cout / ms / 1e9, 1)}), flush=True)
    del xp, yp, ws
