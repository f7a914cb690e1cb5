#!/usr/bin/env python3
"""A/B of the two bf16-PLANE 3x3 convolutions on the CubeNET layer shapes: conv_bf16v2.hip (one persistent 8-wave workgroup per
CU, 32x32x16 MFMA) against conv_bf16v3.hip (two 4-wave workgroups per CU, 16x16x32 MFMA, stores from the accumulators).
Interleaved rounds in one process on the same random data (rule 24); outputs and BN partial statistics compared.
    usage: v3_bench.py [out.json]      env: SHAPES='[[N,H,W,Cin,Cout],...]'  STAGGER='0,3000,6000' (extra v3 arms via _dbg)"""
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
STAGGERS = [int(v) for v in os.environ.get("STAGGER", "").split(",") if v]


def rup(x, m):
    return (x + m - 1) // m * m


def chan_stats(stats, tiles, cout_pad, cout):
    s = stats.view(tiles, cout_pad, 4).double()
    n = s[:, :, 2]
    mean = (s[:, :, 0] * n).sum(0) / n.sum(0)
    m2 = (s[:, :, 1] + n * (s[:, :, 0] - mean) ** 2).sum(0)
    return mean[:cout], (m2 / n.sum(0))[:cout]


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    rows = []
    for (N, H, W, Cin, Cout) in SHAPES:
        cs16, cout_pad = rup(Cin, 32), rup(Cout, 64)
        planes = torch.zeros(N * H * W, cs16, dtype=torch.bfloat16, device=dev)
        data = os.environ.get("DATA", "randn")        # operand statistics move the clock the chip holds (DVFS): say which
        src = torch.randn(N * H * W, Cin, device=dev)
        if data == "relu":
            src = torch.relu(src)                      # what a 3x3 layer inside the networks reads: the output of BN + ReLU
        elif data == "uniform01":
            src = torch.rand(N * H * W, Cin, device=dev)   # the synthetic cube of the first layer: u in [0, 1)
        elif data == "zeros":
            src = torch.zeros(N * H * W, Cin, device=dev)
        planes[:, :Cin] = src.to(torch.bfloat16)
        del src
        w = torch.randn(Cout * Cin * 9, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        flops = 2.0 * N * H * W * Cin * Cout * 9
        wpb = torch.empty(((Cin + 31) // 32) * 9 * cout_pad * 32, dtype=torch.bfloat16, device=dev)
        assert lib.hpri_pack_weight_bf16(P(w), P(wpb), 0, Cin, Cout, cout_pad, 9, Cin, 0, 0, st) == 0
        arms = {}
        for kind in ["v2", "v3"] + [f"v3s{s}" for s in STAGGERS] + (["v3nostats"] if os.environ.get("NOSTATS") else []):
            k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
            plan = lib.hpri_conv_bf16v2_plan if kind == "v2" else lib.hpri_conv_bf16v3_plan
            plan(N, H, W, cs16, cout_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf))
            ws = torch.empty(max(wsf.value, 4), device=dev)
            stats = torch.zeros(tl.value * cout_pad * 4, device=dev)
            y = torch.zeros(N * H * W * Cout, device=dev)
            if kind == "v2":
                def call(y=y, stats=stats, ws=ws):
                    return lib.hpri_conv_bf16v2(P(planes), 0, cs16, 0, P(wpb), P(b), P(y), Cout, 0, P(stats), N, H, W, cs16, Cout,
                                                cout_pad, Cout, 0, 0, P(ws), ws.numel(), st)
            elif kind == "v3nostats":
                def call(y=y, stats=stats, ws=ws):
                    return lib.hpri_conv_bf16v3(P(planes), 0, cs16, 0, P(wpb), P(b), P(y), Cout, 0, P(None), N, H, W, cs16, Cout,
                                                cout_pad, Cout, 0, 0, P(ws), ws.numel(), st)
            elif kind == "v3":
                def call(y=y, stats=stats, ws=ws):
                    return lib.hpri_conv_bf16v3(P(planes), 0, cs16, 0, P(wpb), P(b), P(y), Cout, 0, P(stats), N, H, W, cs16, Cout,
                                                cout_pad, Cout, 0, 0, P(ws), ws.numel(), st)
            else:
                sg = int(kind[3:])

                def call(y=y, stats=stats, ws=ws, sg=sg):
                    return lib.hpri_conv_bf16v3_dbg(P(planes), 0, cs16, 0, P(wpb), P(b), P(y), Cout, 0, P(stats), N, H, W, cs16, Cout,
                                                    cout_pad, Cout, 0, 0, P(ws), ws.numel(), P(None), sg, st)
            arms[kind] = (call, y, stats, k.value, tl.value)
        res = {k: [] for k in arms}
        for rnd in range(6):
            for kind, (fn, *_r) in arms.items():
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
        y2, y3 = arms["v2"][1], arms["v3"][1]
        m2, v2 = chan_stats(arms["v2"][2], arms["v2"][4], cout_pad, Cout)
        m3, v3 = chan_stats(arms["v3"][2], arms["v3"][4], cout_pad, Cout)
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        row = {"shape": [N, H, W, Cin, Cout], "ms": med, "tf": {k: flops / v / 1e9 for k, v in med.items()},
               "ksplit": {k: arms[k][3] for k in arms}, "max_abs_dy": float((y2 - y3).abs().max()),
               "max_abs_y": float(y2.abs().max()), "max_dmean": float((m2 - m3).abs().max()),
               "max_dvar_rel": float(((v2 - v3).abs() / v2).max())}
        rows.append(row)
        print(f"N{N} {H}x{W} {Cin}->{Cout}: " + "  ".join(f"{k} {row['tf'][k]:7.1f} TF ({med[k]:.3f} ms, k{row['ksplit'][k]})" for k in med)
              + f"  max|dy| {row['max_abs_dy']:.2e} of {row['max_abs_y']:.1f}  dmean {row['max_dmean']:.1e} dvar {row['max_dvar_rel']:.1e}",
              flush=True)
        del planes, arms, y2, y3
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
