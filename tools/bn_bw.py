#!/usr/bin/env python3
"""HBM rate of the bf16-mode BatchNorm passes (hpri_bn_apply_relu_x16, hpri_bn_relu_bwd_x16_dy16) on the shapes they meet: bytes the
pass must move / time.  usage: bn_bw.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402


def rup(x, m):
    return (x + m - 1) // m * m


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    for name, (npx, C, groups) in {"C2 64ch 2x608x968": (2 * 608 * 968, 64, 1), "C2 128ch 2x304x484": (2 * 304 * 484, 128, 1),
                                   "C2 512ch 2x76x121": (2 * 76 * 121, 512, 1), "C3 1650ch 608x700": (608 * 700, 1650, 1),
                                   "C3 1650ch 608x700 in a 3328-wide concat": (608 * 700, 1650, 1)}.items():
        cs = rup(C, 8)
        cs16 = rup(C, 32) * (2 if "concat" in name else 1)      # (the planes of a skip half: every other 3.3 KB of a 6.6 KB row)
        x16 = (torch.randn(npx, cs, device=dev)).to(torch.bfloat16)
        dy16 = (torch.randn(npx, cs, device=dev)).to(torch.bfloat16)
        planes = torch.empty(npx * cs16, dtype=torch.bfloat16, device=dev)
        dplanes = torch.empty(npx * cs16, dtype=torch.bfloat16, device=dev)
        scale, shift, mean, invstd = (torch.rand(C, device=dev) + 0.5 for _ in range(4))
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        nblk, cpart = ctypes.c_int(), ctypes.c_int()
        lib.hpri_col_reduce_plan(npx, 1, C, ctypes.byref(nblk), ctypes.byref(cpart))
        ws = torch.empty(2 * (nblk.value * 2 * cpart.value + 2 * C) + 64, device=dev)

        def apply():
            return lib.hpri_bn_apply_relu_x16(P(x16), cs, 0, P(None), cs, 0, P(scale), P(shift), npx, npx, C, cs, 1, P(planes), npx * cs16, cs16, 0, rup(C, 32), 1, st)

        def bwd():
            return lib.hpri_bn_relu_bwd_x16_dy16(P(dy16), cs, 0, P(x16), cs, 0, P(None), cs, 0, P(mean), P(invstd), P(scale), P(shift), P(dg), P(db), 0,
                                                 P(None), 0, P(ws), ws.numel(), npx, npx, C, cs, 1, 1, P(dplanes), npx * cs16, cs16, 0, rup(C, 32), 1, st)
        for tag, fn, nbytes in (("apply  (read x16, write planes)", apply, npx * (cs * 2 + rup(C, 32) * 2)),
                                ("bwd    (2 x read dy16 + x16, write planes)", bwd, npx * (4 * cs * 2 + rup(C, 32) * 2))):
            for _ in range(5):
                assert fn() == 0, lib.hpri_last_error()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            print(f"{name:40s} {tag:44s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
