#!/usr/bin/env python3
"""The north_star's named kernel on its own: CubeNET-64's 238->64 3x3 encoder convolution (models.py:169) on a batch of
two 238x608x968 cubes, forward, in the bf16 plane mode (bf16 NHWC input planes already resident, fp32 output + BatchNorm
partial statistics), and for comparison the exact fp32 kernel.  Prints one JSON object; run it under
``rocprofv3 --kernel-trace --stats`` / ``--pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` for profiles/r02_first_conv.*.

Algorithmic work per launch (batch 2): 2 x 161.36 GFLOP; algorithmic HBM bytes: bf16 input planes 2 x 301.3 MB
(256 padded channels) + fp32 output 2 x 150.7 MB + weights 0.3 MB = 904.3 MB."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402

N, H, W, CIN, COUT = 2, 608, 968, 238, 64
FLOPS = 2.0 * N * H * W * CIN * COUT * 9
PEAK_BF16 = 2500.0      # TFLOP/s, MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def _time(call, reps, settle_s, check):
    """(burst ms, sustained ms): ``reps`` launches timed right after three warm-up launches on a chip that has just been idle
    (what rounds 1-3 reported), and the same after ``settle_s`` seconds of back-to-back launches.  The two differ by ~20 % on
    this layer: the chip raises its clock only over tens of milliseconds of load (profiles/r04_first_conv_ramp.txt: 800 TF over
    the first 10 launches from idle, 847 over 50, 945-994 sustained for 20 s with no droop), and a step of the network keeps
    it loaded, so the sustained figure is the one that describes the kernel inside a step."""
    import time
    for _ in range(3):
        check(call())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    burst = e0.elapsed_time(e1) / reps
    if settle_s <= 0:
        return burst, burst
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(100):
            call()
        torch.cuda.synchronize()
    n = max(reps, 50)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return burst, e0.elapsed_time(e1) / n


def measure(reps=20, modes=("bf16_planes", "fp32"), settle_s=1.0, kind=None):
    """``kind`` = "f16": the same launches on libhyperpri_hip_f16.so (IEEE half planes and weights, v_mfma_f32_16x16x32_f16: the
    form the f16 mode runs; the result keys keep their "bf16" names = "the 16-bit type")."""
    with _lib.using(kind):
        return _measure(reps, modes, settle_s)


def _measure(reps, modes, settle_s):
    lib = _lib.current()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    cs, cs16, cout_pad = 240, 256, 64
    # operand statistics move the clock the chip holds (DVFS: MI355X_MICROARCH.md), so the layer is measured on what it reads in
    # the benched workload: the synthetic cube u in [0, 1) (x = u(1234 + n, .)) and PyTorch-default-bound weights
    # w = (2u - 1) / sqrt(fan_in), fan_in = 238 * 9 (bench.py: synth_init_), both from the device generator
    from hyperpri_amd.engine import synth_fill_
    x = torch.zeros(N * H * W, cs, device=dev)
    xv = torch.empty(N * H * W * CIN, device=dev)
    synth_fill_(xv, 1234, mode=0)
    x[:, :CIN] = xv.view(N * H * W, CIN)
    del xv
    w = torch.empty(COUT * CIN * 9, device=dev)
    synth_fill_(w, 1000, mode=2, scale=1.0 / (CIN * 9) ** 0.5)
    b = torch.empty(COUT, device=dev)
    synth_fill_(b, 1001, mode=2, scale=1.0 / (CIN * 9) ** 0.5)
    y = torch.empty(N * H * W * COUT, device=dev)
    out = {"shape": {"N": N, "H": H, "W": W, "Cin": CIN, "Cout": COUT, "ks": 3}, "gflop_per_launch": FLOPS / 1e9}
    for mode in modes:
        k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        if mode in ("bf16_planes", "bf16_out_only", "f32_out_only"):
            planes = torch.empty(N * H * W * cs16, dtype=torch.bfloat16, device=dev)
            assert lib.hpri_to_planes(P(x), cs, 0, P(planes), 0, cs16, 0, N * H * W, CIN, cs16, 1, st) == 0
            wp = torch.empty((cs16 // 32) * 9 * cout_pad * 32, dtype=torch.bfloat16, device=dev)
            assert lib.hpri_pack_weight_bf16(P(w), P(wp), 0, CIN, COUT, cout_pad, 9, CIN, 0, 0, st) == 0
            kern = "hpri_conv_bf16v3"        # the kernel the bf16 mode runs
            getattr(lib, kern + "_plan")(N, H, W, cs16, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
            stats = torch.empty(tl.value * cout_pad * 4, device=dev)
            ws = torch.empty(max(wsf.value, 4), device=dev)
            conv = getattr(lib, kern)
            call = lambda: conv(P(planes), 0, cs16, 0, P(wp), P(b), P(y), COUT, 0, P(stats), N, H, W, cs16, COUT,
                                cout_pad, COUT, 0, 0, P(ws), ws.numel(), st)
            out["bf16_kernel"] = kern
            alg_bytes = N * H * W * (cs16 * 2 + COUT * 4) + wp.numel() * 2
            if kern == "hpri_conv_bf16v3":
                # the form the bf16 STEP runs (engine.YR_BF16): the pre-BN tensor leaves as bf16 straight from the accumulators
                # (statistics from the fp32 sums) -- SURVEY.md 7.3-2's fp16-in / fp16-out variant of the roofline claim
                y16 = torch.empty(N * H * W * COUT, dtype=torch.bfloat16, device=dev)
                call16 = lambda: conv(P(planes), 0, cs16, 0, P(wp), P(b), P(y16), COUT, 0, P(stats), N, H, W, cs16, COUT,
                                      cout_pad, COUT, 4, 0, P(ws), ws.numel(), st)
                alg16 = N * H * W * (cs16 * 2 + COUT * 2) + wp.numel() * 2
        else:
            wp = torch.empty(lib.hpri_packed_weight_floats(CIN, cout_pad, 9), device=dev)
            assert lib.hpri_pack_weight(P(w), P(wp), 0, CIN, COUT, cout_pad, 9, 0, 0, CIN, st) == 0
            lib.hpri_conv_fwd_plan(N, H, W, cs, cout_pad, 3, 0, 0, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
            stats = torch.empty(tl.value * cout_pad * 4, device=dev)
            ws = torch.empty(max(wsf.value, 4), device=dev)
            call = lambda: lib.hpri_conv_fwd(P(x), cs, 0, P(wp), P(b), P(y), COUT, 0, P(stats), N, H, W, cs, COUT, cout_pad, COUT, 3,
                                             0, 0, 0, 0, 0, 0, 0, 0, P(ws), ws.numel(), st)
            alg_bytes = N * H * W * (cs * 4 + COUT * 4) + wp.numel() * 4
        def check(rc):
            assert rc == 0, lib.hpri_last_error()
        if mode == "bf16_out_only":      # one output form per process: what the counter passes (tools/pmc_cmd.sh) attribute to the kernel
            b16, ms16 = _time(call16, reps, settle_s, check)
            out["bf16_planes_bf16_out"] = {"ms": round(ms16, 4), "tflops": round(FLOPS / ms16 / 1e9, 1),
                                           "frac_of_2.5PF": round(FLOPS / ms16 / 1e9 / PEAK_BF16, 4),
                                           "algorithmic_hbm_mb": round(alg16 / 1e6, 1), "ms_burst_from_idle": round(b16, 4)}
            continue
        burst, ms = _time(call, reps, settle_s, check)
        tf = FLOPS / ms / 1e9
        out[mode] = {"ms": round(ms, 4), "tflops": round(tf, 1), "algorithmic_hbm_mb": round(alg_bytes / 1e6, 1),
                     "algorithmic_tb_s": round(alg_bytes / ms / 1e9, 2), "settle_s": settle_s,
                     "ms_burst_from_idle": round(burst, 4), "tflops_burst_from_idle": round(FLOPS / burst / 1e9, 1)}
        if mode == "f32_out_only":
            out[mode]["frac_of_2.5PF"] = round(tf / PEAK_BF16, 4)
        elif mode == "bf16_planes":
            out[mode]["frac_of_2.5PF"] = round(tf / PEAK_BF16, 4)
            if out.get("bf16_kernel") == "hpri_conv_bf16v3":
                b16, ms16 = _time(call16, reps, settle_s, check)
                out["bf16_planes_bf16_out"] = {"ms": round(ms16, 4), "tflops": round(FLOPS / ms16 / 1e9, 1),
                                               "frac_of_2.5PF": round(FLOPS / ms16 / 1e9 / PEAK_BF16, 4),
                                               "algorithmic_hbm_mb": round(alg16 / 1e6, 1), "algorithmic_tb_s": round(alg16 / ms16 / 1e9, 2),
                                               "ms_burst_from_idle": round(b16, 4)}
        else:
            out[mode]["frac_of_157.3TF"] = round(tf / 157.3, 4)
    # ---- fp32 Winograd F(2x2,3x3) on the same layer (the headline path; conv_wino4.hip) ----
    if "fp32" in modes:
        up = torch.empty(lib.hpri_wino_packed_floats(CIN, cout_pad), device=dev)
        assert lib.hpri_wino4_pack(P(w), P(up), ctypes.c_void_p(0), 0, CIN, COUT, cout_pad, CIN, st) == 0
        tlw = ctypes.c_int()
        lib.hpri_conv_wino4_plan(N, H, W, ctypes.byref(tlw))
        statsw = torch.zeros(tlw.value * cout_pad * 4, device=dev)
        callw = lambda: lib.hpri_conv_wino4(P(x), cs, 0, P(up), P(b), P(y), COUT, 0, P(statsw), N, H, W, cs, COUT, cout_pad, COUT, 0, st)
        for _ in range(3):
            assert callw() == 0, lib.hpri_last_error()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            callw()
        e1.record()
        torch.cuda.synchronize()
        msw = e0.elapsed_time(e1) / reps
        executed = 2.0 * N * ((H + 1) // 2) * ((W + 1) // 2) * 16 * CIN * COUT
        out["fp32_winograd"] = {"ms": round(msw, 4), "executed_tflops": round(executed / msw / 1e9, 1),
                                "direct_equivalent_tflops": round(FLOPS / msw / 1e9, 1),
                                "frac_of_157.3TF_executed": round(executed / msw / 1e9 / 157.3, 4)}
    return out


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    modes = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ("bf16_planes", "fp32")
    print(json.dumps(measure(reps, modes, settle_s=float(os.environ.get("SETTLE", "1.0")))))
