#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md "HBM" prescribes for gfx950: counter unit is KiB; FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads, so it is doubled; WRITE_SIZE is exact.  Writes profiles/<tag>_pmc_traffic.json with
per-launch averages, and the kernel-stats CSV summary next to it.

    python tools/pmc_traffic.py gpurun_out/r01 r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def per_kernel(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: [0.0, set()])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in agg.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fetch = per_kernel(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        rd = 2.0 * f * 1024.0 / max(nf, 1)        # gfx950: FETCH_SIZE counts 128-B requests as 64 B
        wr = w * 1024.0 / max(nw, 1)
        out[k] = {"launches": max(nf, nw), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    if not out:
        raise SystemExit(f"no FETCH_SIZE / WRITE_SIZE rows under {src}: nothing written (bench.py replays profiles/{tag}_pmc_traffic.json)")
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    # the source stamp of the library the passes ran (tools/measure.sh copies it next to the counters; fallback: the local build):
    # bench.py replays these figures only while the loaded library carries the same stamp
    stamp = None
    for cand in (os.path.join(src, "lib_stamp.txt"), os.path.join(root, "hyperpri_amd", "lib", "libhyperpri_hip.so.stamp")):
        if os.path.exists(cand):
            stamp = open(cand).read().strip()
            break
    with open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w") as fh:
        json.dump({"library_stamp": stamp, "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --steps 2 --warmup 1; "
                           "bytes = counter*1024, FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)",
                   "kernels": out}, fh, indent=1)
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(root, "profiles", f"{tag}_bench_kernel_stats.csv"))
    for f in glob.glob(os.path.join(src, "stats1", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(root, "profiles", f"{tag}_bench_kernel_stats_one_stream.csv"))
    for f in ("bench.json", "bench_prof.json"):
        p = os.path.join(src, f)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(root, "profiles", f"{tag}_{f}"))
    # third pass: SQ/GRBM counters -> in-kernel clock and MFMA pipe utilisation per kernel
    sqfiles = glob.glob(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)
    if sqfiles:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in sqfiles:
            seen = set()
            for r in csv.DictReader(open(f)):
                a = agg[r["Kernel_Name"]]
                a[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    a["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    a["n"] += 1
        sq = {}
        for k, a in agg.items():
            if a["ns"] <= 0:
                continue
            sq[k] = {"launches": int(a["n"]), "total_ms": a["ns"] / 1e6,
                     "clock_ghz": a["GRBM_GUI_ACTIVE"] / 8.0 / a["ns"],
                     "mfma_busy_simds_of_4": a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["SQ_BUSY_CU_CYCLES"], 1.0),
                     "wave_wait_any_frac": a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1.0),
                     "wave_wait_inst_frac": a["SQ_WAIT_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1.0),
                     "wave_active_frac": a["SQ_ACTIVE_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1.0)}
        with open(os.path.join(root, "profiles", f"{tag}_pmc_sq.json"), "w") as fh:
            json.dump({"note": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                               "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES on bench.py --steps 2 --warmup 1; clock = "
                               "GRBM_GUI_ACTIVE/8/duration (MI355X_MICROARCH.md DVFS note); mfma_busy_simds_of_4 = "
                               "SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES (4.0 = all four SIMDs' matrix pipes busy)",
                       "kernels": dict(sorted(sq.items(), key=lambda kv: -kv[1]["total_ms"])[:12])}, fh, indent=1)
        for k, v in sorted(sq.items(), key=lambda kv: -kv[1]["total_ms"])[:4]:
            print(f"{k[:50]:50s} clk {v['clock_ghz']:.2f} GHz  mfma busy {v['mfma_busy_simds_of_4']:.2f}/4")
    top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]
    for k, v in top:
        print(f"{k[:60]:60s} n={v['launches']:4d} rd={v['read_bytes_per_launch'] / 1e6:9.1f} MB wr={v['write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
