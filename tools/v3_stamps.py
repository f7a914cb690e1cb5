#!/usr/bin/env python3
"""Where a workgroup of conv_bf16v3 spends its cycles: diagnostic builds (tools/build_v3_diag.sh) with s_memtime stamps and parts of
the main loop removed.  Per variant: kernel time by HIP events, and the median over workgroups of main-loop cycles per stage
(ideal: 2 waves x 48 MFMAs x 16 cycles = 1536 with two workgroups on the CU), cycles between a stage's top and the end of its
barrier, store / statistics epilogue, and the in-kernel clock.  Shares, not run times (stamps perturb the kernel)."""
import os as _os
_os.environ.setdefault("HPRI_DIAG", "1")     # uses entry points of the DIAGNOSTICS build (include/hyperpri_hip_diag.h): HPRI_DIAG=1 python -m hyperpri_amd.build
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402

SHAPES = json.loads(os.environ["SHAPES"]) if "SHAPES" in os.environ else [(2, 608, 968, 238, 64), (2, 304, 484, 256, 128)]
DIAGS = [int(v) for v in os.environ.get("DIAGS", "0,1,2,3,4,8,15").split(",")]
STAGGER = int(os.environ.get("STAGGER", "6000"))


def rup(x, m):
    return (x + m - 1) // m * m


def main():
    main_lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    for (N, H, W, Cin, Cout) in SHAPES:
        cs16, cout_pad = rup(Cin, 32), rup(Cout, 64)
        planes = torch.zeros(N * H * W, cs16, dtype=torch.bfloat16, device=dev)
        planes[:, :Cin] = torch.randn(N * H * W, Cin, device=dev).to(torch.bfloat16)
        w = torch.randn(Cout * Cin * 9, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        wpb = torch.empty((cs16 // 32) * 9 * cout_pad * 32, dtype=torch.bfloat16, device=dev)
        assert main_lib.hpri_pack_weight_bf16(P(w), P(wpb), 0, Cin, Cout, cout_pad, 9, Cin, 0, 0, st) == 0
        k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        main_lib.hpri_conv_bf16v3_plan(N, H, W, cs16, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
        ws = torch.empty(max(wsf.value, 4), device=dev)
        stats = torch.zeros(tl.value * cout_pad * 4, device=dev)
        y = torch.zeros(N * H * W * Cout, device=dev)
        nwg = 4096
        nstages = (cs16 // 32) * 3 // k.value
        flops = 2.0 * N * H * W * Cin * Cout * 9
        print(f"N{N} {H}x{W} {Cin}->{Cout}: {nwg} workgroups, ksplit {k.value}, {nstages} stages per workgroup")
        for d in DIAGS:
            lib = ctypes.CDLL(os.path.join(ROOT, "hyperpri_amd", "lib", f"libv3diag{d}.so"))
            lib.hpri_last_error.restype = ctypes.c_char_p
            stamps = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)

            def call(sp):
                rc = lib.hpri_conv_bf16v3_dbg(P(planes), ctypes.c_longlong(0), cs16, 0, P(wpb), P(b), P(y), Cout, 0, P(stats), N, H, W,
                                              cs16, Cout, cout_pad, Cout, 0, 0, P(ws), ctypes.c_size_t(ws.numel()), P(sp), STAGGER, st)
                assert rc == 0, lib.hpri_last_error()
            for _ in range(3):
                call(None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call(None)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            call(stamps)
            torch.cuda.synchronize()
            t = stamps.view(-1, 16).cpu().double()
            t = t[t[:, 0] > 0]
            loop = t[:, 1] - t[:, 0]                       # first item of every workgroup: prologue wait + main loop
            wait = t[:, 4] / t[:, 9].clamp(min=1)          # top-of-stage wait + barrier, all items
            store = t[:, 2] - t[:, 1]
            stat = t[:, 3] - t[:, 2]
            life = t[:, 8] - t[:, 0]                       # cycles from the first stamp to the end of the workgroup
            real = (t[:, 5] - t[:, 7]) * 10e-9             # s_memrealtime: 100 MHz
            clk = (life / real.clamp(min=1e-9)).median() / 1e9
            per_item = (life / t[:, 9].clamp(min=1)).median()
            print(f"   diag {d:2d}: {ms:.3f} ms = {flops / ms / 1e9:7.1f} TF | first loop {loop.median():7.0f} cyc = {loop.median() / nstages:5.0f}/stage"
                  f" | top wait+barrier {wait.median() / nstages:4.0f}/stage | store {store.median():5.0f} | stats {stat.median():5.0f}"
                  f" | {per_item:7.0f} cyc/item ({per_item / nstages:5.0f}/stage; MFMA-bound with a partner {2 * 48 * 16}) | items/wg {t[:, 9].median():.0f}"
                  f" | clock {clk:.2f} GHz | {len(t)} wgs")
            if (t[:, 10] > 0).any():          # round 4: inside the epilogue of the first item (wave 0): loop end -> next item's prologue issued
                e = t[t[:, 10] > 0]           #          -> transposition done -> stores issued -> statistics written
                print(f"      epilogue of the first item: next-item set-up + prologue DMA {(e[:, 10] - e[:, 1]).median():6.0f} | bias + transposition "
                      f"{(e[:, 11] - e[:, 10]).median():6.0f} | addresses + stores {(e[:, 12] - e[:, 11]).median():6.0f} | statistics {(e[:, 2] - e[:, 12]).median():6.0f}")


if __name__ == "__main__":
    main()
