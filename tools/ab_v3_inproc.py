#!/usr/bin/env python3
"""In-process A/B of whole-library variants on the bf16 plane convolution (cdna_hip_programming.md rule 24: interleaved rounds in ONE
process on one device): every hyperpri_amd/lib/var_<name>.so named on the command line (tools/build_v3_variants.sh) is loaded with
ctypes and hpri_conv_bf16v3 is timed on the CubeNET layer shapes, arms interleaved, after a settle phase under load; outputs of
all arms must be bit-identical to the first.     usage: ab_v3_inproc.py name1 name2 ... [--shapes first|all] [--out16]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402
from hyperpri_amd.engine import synth_fill_  # noqa: E402

SHAPES = {"first": [(2, 608, 968, 238, 64)],
          "all": [(2, 608, 968, 238, 64), (2, 608, 968, 64, 64), (2, 608, 968, 128, 64), (2, 304, 484, 128, 128), (2, 304, 484, 256, 128),
                  (2, 152, 242, 256, 256), (2, 152, 242, 512, 256), (2, 76, 121, 512, 512), (2, 76, 121, 1024, 512), (2, 38, 60, 1024, 1024)]}


def rup(x, m):
    return (x + m - 1) // m * m


def bind(path):
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _lib.parse_header().items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    which = "all" if "--shapes=all" in sys.argv or ("--shapes" in sys.argv and "all" in sys.argv) else "first"
    args = [a for a in args if a not in ("all", "first")]
    out16 = "--out16" in sys.argv
    base = _lib.load()
    libs = {n: bind(os.path.join(ROOT, "hyperpri_amd", "lib", f"var_{n}.so")) for n in args}
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    rows = []
    for (N, H, W, Cin, Cout) in SHAPES[which]:
        cs16, cout_pad = rup(Cin, 32), rup(Cout, 64)
        planes = torch.zeros(N * H * W, cs16, dtype=torch.bfloat16, device=dev)
        xv = torch.empty(N * H * W * Cin, device=dev)
        synth_fill_(xv, 1234, mode=0)
        planes[:, :Cin] = (xv.view(N * H * W, Cin) if Cin == 238 else torch.relu(xv.view(N * H * W, Cin) - 0.5)).to(torch.bfloat16)
        del xv
        w = torch.empty(Cout * Cin * 9, device=dev)
        synth_fill_(w, 1000, mode=2, scale=1.0 / (Cin * 9) ** 0.5)
        b = torch.empty(Cout, device=dev)
        synth_fill_(b, 1001, mode=2, scale=0.01)
        wp = torch.empty((cs16 // 32) * 9 * cout_pad * 32, dtype=torch.bfloat16, device=dev)
        assert base.hpri_pack_weight_bf16(P(w), P(wp), 0, Cin, Cout, cout_pad, 9, Cin, 0, 0, st) == 0
        k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        base.hpri_conv_bf16v3_plan(N, H, W, cs16, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
        ws = torch.empty(max(wsf.value, 4), device=dev)
        use16 = out16 and k.value == 1
        outs, calls = {}, {}
        for n, lib in libs.items():
            y = torch.zeros(N * H * W * Cout, dtype=torch.bfloat16 if use16 else torch.float32, device=dev)
            stats = torch.zeros(tl.value * cout_pad * 4, device=dev)
            outs[n] = (y, stats)
            calls[n] = (lambda lib=lib, y=y, stats=stats: lib.hpri_conv_bf16v3(P(planes), 0, cs16, 0, P(wp), P(b), P(y), Cout, 0, P(stats), N, H, W,
                                                                              cs16, Cout, cout_pad, Cout, 4 if use16 else 0, 0, P(ws), ws.numel(), st))
        flops = 2.0 * N * H * W * Cin * Cout * 9
        for n, fn in calls.items():
            assert fn() == 0, n
        torch.cuda.synchronize()
        first = args[0]
        same = {n: {"y": bool(torch.equal(outs[n][0], outs[first][0])),
                    "stats_max_rel": float(((outs[n][1] - outs[first][1]).abs().max() / outs[first][1].abs().max().clamp_min(1e-30)))} for n in args}
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:          # settle under load
            for fn in calls.values():
                for _ in range(20):
                    fn()
            torch.cuda.synchronize()
        res = {n: [] for n in args}
        for rnd in range(9):
            for n, fn in calls.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[n].append(e0.elapsed_time(e1) / 20)
        row = {"shape": [N, H, W, Cin, Cout], "out": "bf16" if use16 else "f32",
               "tf_median": {n: round(flops / sorted(v)[len(v) // 2] / 1e9, 1) for n, v in res.items()},
               "tf_best": {n: round(flops / min(v) / 1e9, 1) for n, v in res.items()}, "bit_identical_to_first": same}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del planes, outs, calls
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
