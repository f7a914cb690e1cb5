#!/usr/bin/env python3
"""Is the bf16 plane convolution limited by the chip's power management?  The same launch (238->64, 608x968, batch 2) on
all-zero operands (no switching activity in the matrix pipe, LDS and DMA data paths) and on random operands: the instruction
stream and the memory traffic are identical, only the data -- and with it the power draw and the clock the chip holds -- differ."""
import os as _os
_os.environ.setdefault("HPRI_DIAG", "1")     # uses entry points of the DIAGNOSTICS build (include/hyperpri_hip_diag.h): HPRI_DIAG=1 python -m hyperpri_amd.build
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402

N, H, W, CIN, COUT = 2, 608, 968, 238, 64
lib = _lib.load()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
cs16, cout_pad = 256, 64
flops = 2.0 * N * H * W * CIN * COUT * 9
out = {}
for name in ("zeros", "random", "zeros", "random"):
    planes = torch.zeros(N * H * W * cs16, dtype=torch.bfloat16, device=dev)
    wp = torch.zeros((cs16 // 32) * 9 * cout_pad * 32, dtype=torch.bfloat16, device=dev)
    if name == "random":
        planes.copy_(torch.randn(N * H * W * cs16, device=dev).to(torch.bfloat16))
        wp.copy_((torch.randn(wp.numel(), device=dev) * 0.05).to(torch.bfloat16))
    b = torch.zeros(COUT, device=dev)
    y = torch.empty(N * H * W * COUT, device=dev)
    k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    lib.hpri_conv_bf16v2_plan(N, H, W, cs16, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
    stats = torch.empty(tl.value * cout_pad * 4, device=dev)
    ws = torch.empty(max(wsf.value, 4), device=dev)
    call = lambda: lib.hpri_conv_bf16v2(P(planes), 0, cs16, 0, P(wp), P(b), P(y), COUT, 0, P(stats), N, H, W, cs16, COUT, cout_pad, COUT, 0, 0,
                                        P(ws), ws.numel(), st)
    for _ in range(20):
        assert call() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    out.setdefault(name, []).append(round(flops / ms / 1e9, 1))
    print(f"{name:7s}: {ms:.4f} ms = {flops / ms / 1e9:.0f} TF", flush=True)
print(json.dumps(out))

# ---- the same question for the fp32 Winograd kernel (conv_wino4.hip) on the same layer ----
cs = 240
executed = 2.0 * N * ((H + 1) // 2) * ((W + 1) // 2) * 16 * CIN * COUT
outw = {}
for name in ("zeros", "random", "zeros", "random"):
    x = torch.zeros(N * H * W, cs, device=dev)
    w = torch.zeros(COUT, CIN, 3, 3, device=dev)
    if name == "random":
        x[:, :CIN] = torch.randn(N * H * W, CIN, device=dev)
        w.copy_(torch.randn(COUT, CIN, 3, 3, device=dev) * 0.05)
    up = torch.empty(lib.hpri_wino_packed_floats(CIN, cout_pad), device=dev)
    assert lib.hpri_wino4_pack(P(w), P(up), ctypes.c_void_p(0), 0, CIN, COUT, cout_pad, CIN, st) == 0
    tlw = ctypes.c_int()
    lib.hpri_conv_wino4_plan(N, H, W, ctypes.byref(tlw))
    statsw = torch.zeros(tlw.value * cout_pad * 4, device=dev)
    b = torch.zeros(COUT, device=dev)
    y = torch.empty(N * H * W * COUT, device=dev)
    callw = lambda: lib.hpri_conv_wino4(P(x), cs, 0, P(up), P(b), P(y), COUT, 0, P(statsw), N, H, W, cs, COUT, cout_pad, COUT, 0, st)
    for _ in range(10):
        assert callw() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        callw()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    outw.setdefault(name, []).append(round(executed / ms / 1e9, 1))
    print(f"fp32 Winograd {name:7s}: {ms:.4f} ms = {executed / ms / 1e9:.1f} executed TF", flush=True)
print(json.dumps({"fp32_winograd_executed_tf": outw}))
