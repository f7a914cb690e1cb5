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
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    with open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w") as fh:
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --steps 2 --warmup 1; "
                           "bytes = counter*1024, FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)",
                   "kernels": out}, fh, indent=1)
    for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(root, "profiles", f"{tag}_bench_kernel_stats.csv"))
    for f in ("bench.json", "bench_prof.json"):
        p = os.path.join(src, f)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(root, "profiles", f"{tag}_{f}"))
    top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]
    for k, v in top:
        print(f"{k[:60]:60s} n={v['launches']:4d} rd={v['read_bytes_per_launch'] / 1e6:9.1f} MB wr={v['write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
