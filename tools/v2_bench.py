#!/usr/bin/env python3
"""A/B of the round-1 bf16 convolution (fp32 activations, VGPR staging) against the bf16-PLANE kernel
(conv_bf16v2.hip, LDS-DMA for both operands) on the CubeNET layer shapes: interleaved rounds in one process on the
same random data (rule 24), outputs and BN partial statistics compared.   usage: v2_bench.py [out.json]"""
import os as _os
_os.environ.setdefault("HPRI_DIAG", "1")     # uses entry points of the DIAGNOSTICS build (include/hyperpri_hip_diag.h): HPRI_DIAG=1 python -m hyperpri_amd.build
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hyperpri_amd import _lib  # noqa: E402

SHAPES = json.loads(os.environ["SHAPES"]) if "SHAPES" in os.environ else [  # N, H, W, Cin, Cout
    (2, 608, 968, 238, 64), (2, 608, 968, 64, 64), (2, 608, 968, 128, 64), (2, 304, 484, 64, 128),
    (2, 304, 484, 128, 128), (2, 304, 484, 256, 128), (2, 152, 242, 256, 256), (2, 152, 242, 512, 256),
    (2, 76, 121, 512, 512), (2, 76, 121, 1024, 512), (2, 38, 60, 1024, 1024),
]


def rup(x, m):
    return (x + m - 1) // m * m


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rows = []
    for (N, H, W, Cin, Cout) in SHAPES:
        cs = rup(Cin, 8)
        cs16 = rup(Cin, 32)
        x = torch.zeros(N * H * W, cs, device=dev)
        x[:, :Cin] = torch.randn(N * H * W, Cin, device=dev)
        w = torch.randn(Cout * Cin * 9, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        cout_pad = rup(Cout, 64)
        flops = 2.0 * N * H * W * Cin * Cout * 9
        planes = torch.empty(N * H * W * cs16, dtype=torch.bfloat16, device=dev)
        wpb = torch.empty(((Cin + 31) // 32) * 9 * cout_pad * 32, dtype=torch.bfloat16, device=dev)
        assert lib.hpri_pack_weight_bf16(P(w), P(wpb), 0, Cin, Cout, cout_pad, 9, Cin, 0, 0, st) == 0
        ys, sts = [], []
        calls = {}
        for kind in ("old", "v2"):
            k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
            if kind == "old":
                lib.hpri_conv_fwd_bf16_plan(N, H, W, cs, cout_pad, 3, 0, 0, 0, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
            else:
                lib.hpri_conv_bf16v2_plan(N, H, W, cs16, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
            ws = torch.empty(max(wsf.value, 4), device=dev)
            stats = torch.zeros(tl.value * cout_pad * 4, device=dev)
            y = torch.zeros(N * H * W * Cout, device=dev)
            if kind == "old":
                def call(y=y, stats=stats, ws=ws):
                    return lib.hpri_conv_fwd_bf16(P(x), cs, 0, P(wpb), P(b), P(y), Cout, 0, P(stats), N, H, W, cs, Cout, cout_pad,
                                                  Cout, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, P(ws), ws.numel(), st)
            else:
                def call(y=y, stats=stats, ws=ws):
                    return lib.hpri_conv_bf16v2(P(planes), 0, cs16, 0, P(wpb), P(b), P(y), Cout, 0, P(stats), N, H, W, cs16, Cout,
                                                cout_pad, Cout, 0, 0, P(ws), ws.numel(), st)
            calls[kind] = (call, y, stats, k.value, tl.value)

        def to_planes():
            return lib.hpri_to_planes(P(x), cs, 0, P(planes), 0, cs16, 0, N * H * W, Cin, cs16, 1, st)
        assert to_planes() == 0
        res = {"old": [], "v2": [], "to_planes": []}
        for rnd in range(6):
            for kind, fn in (("old", calls["old"][0]), ("v2", calls["v2"][0]), ("to_planes", to_planes)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 5
                e0.record()
                for _ in range(reps):
                    rc = fn()
                    assert rc == 0, lib.hpri_last_error()
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    res[kind].append(e0.elapsed_time(e1) / reps)
        yo, yn = calls["old"][1], calls["v2"][1]
        # per-channel mean / variance from the per-tile (mean, M2, count) records must agree (tile shapes differ)
        def chan_stats(stats, tiles):
            s = stats.view(tiles, cout_pad, 4).double()
            n = s[:, :, 2]
            mean = (s[:, :, 0] * n).sum(0) / n.sum(0)
            m2 = (s[:, :, 1] + n * (s[:, :, 0] - mean) ** 2).sum(0)
            return mean[:Cout], (m2 / n.sum(0))[:Cout]
        mo, vo = chan_stats(calls["old"][2], calls["old"][4])
        mn, vn = chan_stats(calls["v2"][2], calls["v2"][4])
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        row = {"shape": [N, H, W, Cin, Cout], "old_ms": med["old"], "v2_ms": med["v2"], "to_planes_ms": med["to_planes"],
               "old_tf": flops / med["old"] / 1e9, "v2_tf": flops / med["v2"] / 1e9, "ksplit_old": calls["old"][3],
               "ksplit_v2": calls["v2"][3], "max_abs_dy": float((yo - yn).abs().max()), "max_abs_y": float(yo.abs().max()),
               "max_dmean": float((mo - mn).abs().max()), "max_dvar_rel": float(((vo - vn).abs() / vo).max())}
        rows.append(row)
        print(f"N{N} {H}x{W} {Cin}->{Cout}: old {row['old_tf']:7.1f} TF ({med['old']:.3f} ms, k{row['ksplit_old']})  "
              f"v2 {row['v2_tf']:7.1f} TF ({med['v2']:.3f} ms, k{row['ksplit_v2']})  to_planes {med['to_planes']:.3f} ms  "
              f"max|dy| {row['max_abs_dy']:.2e} of {row['max_abs_y']:.1f}  dmean {row['max_dmean']:.1e} dvar {row['max_dvar_rel']:.1e}",
              flush=True)
        del x, planes, yo, yn, calls
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
