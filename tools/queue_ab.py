#!/usr/bin/env python3
"""Item queue against fixed item lists on the bf16 plane convolution (hpri_conv_bf16v3), layer by layer: the same launch on a
stream that has a queue registered (hpri_set_item_queue) and on one that has none, arms interleaved in one process after a
settle phase; outputs must be bit-identical.     usage: queue_ab.py [--out16] > profiles/r05_queue_ab.jsonl"""
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

SHAPES = [(2, 608, 968, 238, 64), (2, 608, 968, 64, 64), (2, 608, 968, 128, 64), (2, 304, 484, 128, 128), (2, 304, 484, 256, 128),
          (2, 152, 242, 256, 256), (2, 152, 242, 512, 256), (2, 76, 121, 512, 512), (2, 76, 121, 1024, 512), (2, 38, 60, 1024, 1024)]


def rup(x, m):
    return (x + m - 1) // m * m


def main():
    out16 = "--out16" in sys.argv
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    streams = {"fixed": torch.cuda.Stream(device=dev), "queue": torch.cuda.Stream(device=dev)}
    qbuf = torch.zeros(lib.hpri_item_queue_bytes() // 4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    assert lib.hpri_set_item_queue(P(qbuf), qbuf.numel() * 4, ctypes.c_void_p(streams["queue"].cuda_stream)) == 0
    st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (N, H, W, Cin, Cout) in SHAPES:
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
        assert lib.hpri_pack_weight_bf16(P(w), P(wp), 0, Cin, Cout, cout_pad, 9, Cin, 0, 0, st0) == 0
        k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        lib.hpri_conv_bf16v3_plan(N, H, W, cs16, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
        ws = torch.empty(max(wsf.value, 4), device=dev)
        use16 = out16 and k.value == 1
        torch.cuda.synchronize()
        outs, calls = {}, {}
        for n, s in streams.items():
            y = torch.zeros(N * H * W * Cout, dtype=torch.bfloat16 if use16 else torch.float32, device=dev)
            stats = torch.zeros(tl.value * cout_pad * 4, device=dev)
            outs[n] = (y, stats)
            h = ctypes.c_void_p(s.cuda_stream)
            calls[n] = (lambda y=y, stats=stats, h=h: lib.hpri_conv_bf16v3(P(planes), 0, cs16, 0, P(wp), P(b), P(y), Cout, 0, P(stats), N, H, W,
                                                                          cs16, Cout, cout_pad, Cout, 4 if use16 else 0, 0, P(ws), ws.numel(), h))
        torch.cuda.synchronize()
        flops = 2.0 * N * H * W * Cin * Cout * 9
        for n, fn in calls.items():
            assert fn() == 0, n
            torch.cuda.synchronize()
        same = bool(torch.equal(outs["fixed"][0], outs["queue"][0]))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:          # settle under load
            for n, fn in calls.items():
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
        res = {n: [] for n in calls}
        for rnd in range(9):
            for n, fn in calls.items():
                s = streams[n]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(20):
                    fn()
                e1.record(s)
                torch.cuda.synchronize()
                res[n].append(e0.elapsed_time(e1) / 20)
        med = {n: sorted(v)[len(v) // 2] for n, v in res.items()}
        row = {"shape": [N, H, W, Cin, Cout], "ksplit": k.value, "out": "bf16" if use16 else "f32", "us_median": {n: round(v * 1e3, 1) for n, v in med.items()},
               "tf_median": {n: round(flops / v / 1e9, 1) for n, v in med.items()}, "queue_over_fixed_time": round(med["queue"] / med["fixed"], 4),
               "bit_identical": same}
        print(json.dumps(row), flush=True)
        del planes, outs, calls
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
