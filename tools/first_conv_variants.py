#!/usr/bin/env python3
"""What moves the 238->64 bf16-plane convolution (models.py:169): output type (fp32 / bf16 pre-BN tensor, the form the bf16 step
runs), statistics on / off, operand data, burst against sustained (the chip's clock under load: MI355X_MICROARCH.md DVFS).
Interleaved arms in one process (rule 24), then a soak of SOAK seconds of back-to-back launches per arm with the rate of the last
second.    usage: first_conv_variants.py [out.json]      env: SOAK=8"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402
from hyperpri_amd.engine import synth_fill_  # noqa: E402

N, H, W, CIN, COUT = 2, 608, 968, 238, 64
FLOPS = 2.0 * N * H * W * CIN * COUT * 9


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    cs16, cout_pad = 256, 64
    w = torch.empty(COUT * CIN * 9, device=dev)
    synth_fill_(w, 1000, mode=2, scale=1.0 / (CIN * 9) ** 0.5)
    b = torch.empty(COUT, device=dev)
    synth_fill_(b, 1001, mode=2, scale=1.0 / (CIN * 9) ** 0.5)
    wp = torch.empty((cs16 // 32) * 9 * cout_pad * 32, dtype=torch.bfloat16, device=dev)
    assert lib.hpri_pack_weight_bf16(P(w), P(wp), 0, CIN, COUT, cout_pad, 9, CIN, 0, 0, st) == 0
    k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    lib.hpri_conv_bf16v3_plan(N, H, W, cs16, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
    stats = torch.empty(tl.value * cout_pad * 4, device=dev)
    planes = {}
    for data in ("uniform01", "zeros"):
        pl = torch.zeros(N * H * W, cs16, dtype=torch.bfloat16, device=dev)
        if data == "uniform01":
            xv = torch.empty(N * H * W * CIN, device=dev)
            synth_fill_(xv, 1234, mode=0)
            pl[:, :CIN] = xv.view(N * H * W, CIN).to(torch.bfloat16)
            del xv
        planes[data] = pl
    y32 = torch.empty(N * H * W * COUT, device=dev)
    y16 = torch.empty(N * H * W * COUT, dtype=torch.bfloat16, device=dev)

    def arm(data, out16, with_stats):
        pl = planes[data]
        y = y16 if out16 else y32
        return lambda: lib.hpri_conv_bf16v3(P(pl), 0, cs16, 0, P(wp), P(b), P(y), COUT, 0, P(stats if with_stats else None), N, H, W,
                                            cs16, COUT, cout_pad, COUT, 4 if out16 else 0, 0, P(None), 0, st)
    arms = {"f32out+stats": arm("uniform01", False, True), "bf16out+stats": arm("uniform01", True, True),
            "f32out": arm("uniform01", False, False), "bf16out": arm("uniform01", True, False),
            "zeros f32out+stats": arm("zeros", False, True), "zeros bf16out+stats": arm("zeros", True, True)}
    res = {kk: [] for kk in arms}
    for rnd in range(5):
        for kk, fn in arms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                assert fn() == 0, lib.hpri_last_error()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res[kk].append(e0.elapsed_time(e1) / 5)
    out = {"burst": {}, "sustained": {}}
    for kk, v in res.items():
        ms = sorted(v)[len(v) // 2]
        out["burst"][kk] = {"ms": round(ms, 4), "tflops": round(FLOPS / ms / 1e9, 1), "frac_of_2.5PF": round(FLOPS / ms / 1e9 / 2500, 4)}
        print("burst    ", kk, out["burst"][kk], flush=True)
    soak = float(os.environ.get("SOAK", "8"))
    for kk in ("f32out+stats", "bf16out+stats"):
        fn = arms[kk]
        t0 = time.perf_counter()
        rates = []
        while time.perf_counter() - t0 < soak:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                fn()
            e1.record()
            torch.cuda.synchronize()
            rates.append(e0.elapsed_time(e1) / 200)
        ms = rates[-1]
        out["sustained"][kk] = {"seconds": soak, "ms_first": round(rates[0], 4), "ms_last": round(ms, 4), "tflops_last": round(FLOPS / ms / 1e9, 1),
                                "frac_of_2.5PF_last": round(FLOPS / ms / 1e9 / 2500, 4)}
        print("sustained", kk, out["sustained"][kk], flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
