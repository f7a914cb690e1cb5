#!/usr/bin/env python3
"""Per-LAYER HBM traffic of the fp32 Winograd forward / data-gradient kernel (conv_wino4) in one C2 step.

  workload mode (run it under rocprofv3 --kernel-trace --pmc FETCH_SIZE, and again with --pmc WRITE_SIZE, one stream):
      python tools/wino4_layer_traffic.py run <order.json>
    runs 2 warm-up steps and ONE measured step of CubeNET-64 (batch 2, 608x968x238) and writes the shapes of the conv_wino4
    launches of the whole process in launch order.
  parse mode:
      python tools/wino4_layer_traffic.py parse <order.json> <fetch_dir> <write_dir> <out.json>
    matches the conv_wino4 dispatches of the two counter passes (sorted by dispatch id) with the launch order, keeps the last
    step's 35 launches, applies the gfx950 FETCH_SIZE correction (x2, MI355X_MICROARCH.md HBM section; counter unit KiB) and
    prints measured / algorithmic bytes per layer (algorithmic = input view + output + packed U, each once)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(order_path):
    os.environ["HPRI_SIDE_STREAM"] = "0"
    import torch
    import bench
    import hyperpri_amd as HP
    from hyperpri_amd import _lib, engine
    order = []
    real_call = _lib.call

    def spy(name, *args):
        if name == "hpri_conv_wino4":
            # (x, x_cs, x_coff, up, bias, y, y_cs, y_coff, stats, N, H, W, Cin_pad, Cout, Cout_pad, y_cw, accumulate, stream)
            order.append({"x_cs": args[1], "N": args[9], "H": args[10], "W": args[11], "Cin_pad": args[12], "Cout": args[13],
                          "Cout_pad": args[14], "y_cs": args[6], "accumulate": args[16] & 1, "stats": bool(args[8].value)})
        return real_call(name, *args)
    _lib.call = spy
    engine._lib.call = spy
    dev = torch.device("cuda", 0)
    net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
    bench.synth_init_(net)
    x = engine.synth_fill_(torch.empty((2, 1, 238, 608, 968), device=dev), 1234)
    mask = engine.synth_fill_(torch.empty((2, 1, 608, 968), device=dev), 4321, mode=1, thr=0.9)
    crit = torch.nn.BCEWithLogitsLoss()
    for _ in range(3):
        for p in net.parameters():
            p.grad = None
        crit(net(x), mask).backward()
        torch.cuda.synchronize()
    json.dump(order, open(order_path, "w"))
    print("launches", len(order))


def counters(dirname, name):
    rows = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "conv_wino4_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows]


def parse(order_path, fetch_dir, write_dir, out_path):
    order = json.load(open(order_path))
    fetch, write = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
    assert len(fetch) == len(order) == len(write), (len(fetch), len(write), len(order))
    per_step = len(order) // 3
    rows = []
    tot_m = tot_a = 0.0
    for i in range(len(order) - per_step, len(order)):
        o = order[i]
        px = o["N"] * o["H"] * o["W"]
        rd_alg = px * o["Cin_pad"] * 4 + (o["Cin_pad"] // 8) * 16 * 8 * o["Cout_pad"] * 4 + (px * o["Cout"] * 4 if o["accumulate"] else 0)
        wr_alg = px * o["Cout"] * 4
        rd, wr = 2.0 * fetch[i] * 1024.0, write[i] * 1024.0
        rows.append({**o, "read_mb": rd / 1e6, "write_mb": wr / 1e6, "read_alg_mb": rd_alg / 1e6, "write_alg_mb": wr_alg / 1e6,
                     "read_ratio": rd / rd_alg, "total_ratio": (rd + wr) / (rd_alg + wr_alg), "channel_blocks": o["Cout_pad"] // 64})
        tot_m += rd + wr
        tot_a += rd_alg + wr_alg
    json.dump({"note": "conv_wino4 per launch of one C2 step (one stream); FETCH_SIZE x2 x1024, WRITE_SIZE x1024; algorithmic = input "
                       "view + packed U (+ old output when accumulating) read once, output written once",
               "total_ratio": tot_m / tot_a, "layers": rows}, open(out_path, "w"), indent=1)
    print(f"{'N HxW K->N':28s} {'blocks':>6s} {'read MB':>9s} {'alg':>8s} {'ratio':>6s} {'write MB':>9s} {'alg':>8s}")
    for r in rows:
        print(f"{r['N']} {r['H']}x{r['W']} {r['Cin_pad']}->{r['Cout']:<5d}{'+acc' if r['accumulate'] else '    '} {r['channel_blocks']:6d} "
              f"{r['read_mb']:9.1f} {r['read_alg_mb']:8.1f} {r['read_ratio']:6.2f} {r['write_mb']:9.1f} {r['write_alg_mb']:8.1f}")
    print("total measured / algorithmic:", round(tot_m / tot_a, 3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        parse(*sys.argv[2:6])
