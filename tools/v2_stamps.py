#!/usr/bin/env python3
"""Where a workgroup of conv_bf16v2 spends its cycles: diagnostic build with s_memtime stamps (tools/build_stamps.sh).
Prints the median over workgroups of: prologue (first DMA -> first stage visible), main loop, store epilogue,
statistics epilogue, for the early (waves 0-3) and late (waves 4-7) wave group.  Shares, not run times (stamps
perturb the kernel)."""
import os as _os
_os.environ.setdefault("HPRI_DIAG", "1")     # uses entry points of the DIAGNOSTICS build (include/hyperpri_hip_diag.h): HPRI_DIAG=1 python -m hyperpri_amd.build
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SHAPES = json.loads(os.environ["SHAPES"]) if "SHAPES" in os.environ else [(2, 608, 968, 238, 64), (2, 608, 968, 64, 64), (2, 304, 484, 128, 128)]


def rup(x, m):
    return (x + m - 1) // m * m


def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "hyperpri_amd", "lib", "libv2stamps.so"))
    lib.hpri_last_error.restype = ctypes.c_char_p
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    for (N, H, W, Cin, Cout) in SHAPES:
        cs16, cout_pad = rup(Cin, 32), rup(Cout, 64)
        planes = (torch.randn(N * H * W, cs16, device=dev)).to(torch.bfloat16)
        w = torch.randn(Cout * Cin * 9, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        wpb = torch.empty((cs16 // 32) * 9 * cout_pad * 32, dtype=torch.bfloat16, device=dev)
        assert lib.hpri_pack_weight_bf16(P(w), P(wpb), 0, Cin, Cout, cout_pad, 9, Cin, 0, 0, st) == 0
        k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        lib.hpri_conv_bf16v2_plan(N, H, W, cs16, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
        ws = torch.empty(max(wsf.value, 4), device=dev)
        stats = torch.zeros(tl.value * cout_pad * 4, device=dev)
        y = torch.zeros(N * H * W * Cout, device=dev)
        nwg = 8 * ((tl.value + 7) // 8) * (cout_pad // (128 if cout_pad % 128 == 0 else 64)) * k.value + 64
        stamps = torch.zeros(nwg * 2 * 8, dtype=torch.int64, device=dev)
        for _ in range(3):
            rc = lib.hpri_conv_bf16v2_dbg(P(planes), ctypes.c_longlong(0), cs16, 0, P(wpb), P(b), P(y), Cout, 0, P(stats), N, H, W, cs16,
                                          Cout, cout_pad, Cout, 0, 0, P(ws), ctypes.c_size_t(ws.numel()), P(stamps), st)
            assert rc == 0, lib.hpri_last_error()
        torch.cuda.synchronize()
        t = stamps.view(-1, 2, 8).cpu().double()
        live = t[:, 0, 0] > 0
        t = t[live]
        names = ["wait+barrier", "main loop", "next prologue issue", "bias+store", "statistics"]
        print(f"N{N} {H}x{W} {Cin}->{Cout}: {int(live.sum())} workgroups, ksplit {k.value}")
        for g, gname in ((0, "waves 0-3"), (1, "waves 4-7")):
            order = [0, 1, 2, 5, 3, 4]        # stamp ids in program order (last item of every workgroup)
            d = [(t[:, g, order[i + 1]] - t[:, g, order[i]]) for i in range(5)]
            tot = t[:, g, 4] - t[:, g, 0]
            print(f"   {gname}: total {tot.median():8.0f} cyc | " + " | ".join(f"{n} {x.median():7.0f} ({100 * x.median() / tot.median():4.1f} %)" for n, x in zip(names, d)))
        span = t[:, :, 4].max() - t[:, :, 0].min()
        print(f"   kernel span {span:.0f} s_memtime ticks")


if __name__ == "__main__":
    main()
