#!/usr/bin/env python3
"""Launch times of gemm_bf16v3 on the shapes of config C3 (SpectralUNET-1650 at 608x700) and of the ConvTranspose2d layers of
C2 / C5.  usage: python tools/gemm_v3_bench.py [reps]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hyperpri_amd import _lib

lib = _lib.load()
DEV = "cuda:0"
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rup = lambda x, m: (x + m - 1) // m * m
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def linear(N, HW, K, C, stats=True, y16=False):
    kp, cp, cw = rup(K, 32), rup(C, 64), rup(C, 4)
    xp = (torch.rand(N * HW, kp, device=DEV) - 0.5).to(torch.bfloat16)
    wp = (torch.rand((kp // 32) * cp * 32, device=DEV) - 0.5).to(torch.bfloat16)
    b = torch.zeros(C, device=DEV)
    y = torch.empty(N * HW, cw, device=DEV) if not y16 else None
    y2 = torch.empty(N * HW, cw, dtype=torch.bfloat16, device=DEV) if y16 else None
    tl = ctypes.c_int(); lib.hpri_gemm_bf16v3_plan(N, HW, ctypes.byref(tl))
    s = torch.empty(tl.value * cp * 4, device=DEV) if stats else None

    def fn():
        rc = lib.hpri_gemm_bf16v3(P(xp), kp, 0, P(wp), P(b), P(y), cw, 0, P(y2), cw, 0, P(s), cp, N, HW, kp, C, cp, cw, 0, st())
        assert rc == 0, lib.hpri_last_error()
    ms = timeit(fn)
    fl = 2.0 * N * HW * K * C
    by = N * HW * (kp * 2 + cw * (2 if y16 else 4)) + kp * cp * 2
    return {"op": "linear", "N": N, "HW": HW, "K": K, "C": C, "out": "bf16" if y16 else "f32", "ms": round(ms, 4), "TF": round(fl / ms / 1e9, 1),
            "algorithmic_GB": round(by / 1e9, 3), "algorithmic_TB_s": round(by / ms / 1e9, 2)}


def convt(N, H, W, Cin, Cup, planes_only=False):
    kp, ncp = rup(Cin, 32), rup(4 * Cup, 64)
    xp = (torch.rand(N * H * W, kp, device=DEV) - 0.5).to(torch.bfloat16)
    wp = (torch.rand((kp // 32) * ncp * 32, device=DEV) - 0.5).to(torch.bfloat16)
    b = torch.zeros(Cup, device=DEV)
    y = None if planes_only else torch.empty(N * 4 * H * W, 2 * Cup, device=DEV)
    y16 = torch.empty(N * 4 * H * W, 2 * Cup, dtype=torch.bfloat16, device=DEV) if planes_only else None

    def fn():
        rc = lib.hpri_convt_fwd_bf16v3(P(xp), kp, 0, P(wp), P(b), P(y), 2 * Cup, Cup, P(y16), 2 * Cup, Cup, N, H, W, kp, Cup, ncp, 2 * H, 2 * W, 0, 0, st())
        assert rc == 0, lib.hpri_last_error()
    ms = timeit(fn)
    fl = 2.0 * N * H * W * Cin * 4 * Cup
    res = {"op": "convt_fwd", "N": N, "H": H, "W": W, "Cin": Cin, "Cup": Cup, "out": "bf16" if planes_only else "f32", "ms": round(ms, 4), "TF": round(fl / ms / 1e9, 1)}
    dyp = (torch.rand(N * 4 * H * W, Cup, device=DEV) - 0.5).to(torch.bfloat16)
    cinp, dcw = rup(Cin, 64), rup(Cin, 4)
    wpd = (torch.rand((4 * Cup // 32) * cinp * 32, device=DEV) - 0.5).to(torch.bfloat16)
    dx = torch.empty(N * H * W, dcw, device=DEV)

    def fd():
        rc = lib.hpri_convt_dgrad_bf16v3(P(dyp), Cup, 0, P(wpd), P(dx), dcw, 0, N, H, W, Cup, Cin, cinp, dcw, 2 * H, 2 * W, 0, 0, 0, st())
        assert rc == 0, lib.hpri_last_error()
    msd = timeit(fd)
    res.update({"dgrad_ms": round(msd, 4), "dgrad_TF": round(fl / msd / 1e9, 1)})
    return res


def wgrad(Ppx, Cin, Cout):
    kx, ky = rup(Cin, 32), rup(Cout, 32)
    xp = (torch.rand(Ppx, kx, device=DEV) - 0.5).to(torch.bfloat16)
    yp = (torch.rand(Ppx, ky, device=DEV) - 0.5).to(torch.bfloat16)
    sp, cr, nr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.hpri_wgrad1x1_bf16v3_plan(Ppx, kx, rup(Cout, 64), ctypes.byref(sp), ctypes.byref(cr), ctypes.byref(nr))
    ws = torch.empty(sp.value * cr.value * nr.value, device=DEV)
    dw = torch.empty(Cout, Cin, device=DEV)

    def fn():
        rc = lib.hpri_wgrad1x1_bf16v3(P(xp), kx, 0, rup(Cin, 8), P(yp), ky, 0, rup(Cout, 8), P(ws), ws.numel(), Ppx, kx, rup(Cout, 64), st())
        assert rc == 0, lib.hpri_last_error()

    def fr():
        rc = lib.hpri_wgrad_reduce_ex(P(ws), P(dw), sp.value, cr.value, nr.value, Cin, Cout, 1, 0, 0, 0, st())
        assert rc == 0, lib.hpri_last_error()
    ms, msr = timeit(fn), timeit(fr)
    fl = 2.0 * Ppx * Cin * Cout
    return {"op": "wgrad1x1", "P": Ppx, "Cin": Cin, "Cout": Cout, "splits": sp.value, "ms": round(ms, 4), "TF": round(fl / ms / 1e9, 1), "reduce_ms": round(msr, 4)}


if os.environ.get("ONLY") == "wgrad":
    for r in [wgrad(608 * 700, 1650, 1650), wgrad(608 * 700, 3300, 1650), wgrad(608 * 700, 238, 1650), wgrad(2 * 608 * 968, 64, 64)]:
        print(json.dumps(r), flush=True)
    sys.exit(0)

for r in [linear(1, 608 * 700, 238, 1650), linear(1, 608 * 700, 1650, 1650), linear(1, 608 * 700, 1650, 1650, y16=True), linear(1, 608 * 700, 3300, 1650),
          linear(1, 608 * 700, 1650, 1650, stats=False), linear(2, 608 * 968, 64, 64),
          convt(2, 304, 484, 128, 64), convt(2, 152, 242, 256, 128), convt(2, 76, 121, 512, 256), convt(2, 38, 60, 1024, 512),
          convt(2, 304, 484, 128, 64, planes_only=True), convt(2, 304, 484, 256, 128)]:
    print(json.dumps(r), flush=True)
