#!/usr/bin/env python3
"""Summarise the PMC passes of tools/pmc_cmd.sh per kernel:  python tools/pmc_parse.py gpurun_out/TAG [name-filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "conv_"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add((f, r["Dispatch_Id"]))
for f in glob.glob(d + "/pmc_sq/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    if flt not in k:
        continue
    n = len({x[1] for x in disp[k] if "pmc_sq" in x[0]}) or 1
    ms = dur[k] / n
    cyc = v["GRBM_GUI_ACTIVE"] / 8 / n if v.get("GRBM_GUI_ACTIVE") else 0
    print(f"{k[:70]:70s} n={n:3d} avg {ms:8.3f} ms clk {cyc / (ms * 1e6) if ms else 0:.2f} GHz")
    print(f"    mfma_busy_simds {v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(v['SQ_BUSY_CU_CYCLES'], 1):.2f}/4  wait_any {v['SQ_WAIT_ANY'] / max(v['SQ_WAVE_CYCLES'], 1):.2f}"
          f"  wait_inst {v['SQ_WAIT_INST_ANY'] / max(v['SQ_WAVE_CYCLES'], 1):.2f}  active {v['SQ_ACTIVE_INST_ANY'] / max(v['SQ_WAVE_CYCLES'], 1):.2f}")
    if v.get("SQ_LDS_IDX_ACTIVE"):
        print(f"    lds_active/CU-cycle {v['SQ_LDS_IDX_ACTIVE'] / n / 256 / max(cyc, 1):.2f}  bank_conflict/lds_active {v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1):.2f}"
              f"  wait_lds {v['SQ_WAIT_INST_LDS'] / n:.3g}  vmem_cycles {v['SQ_INST_CYCLES_VMEM'] / n:.3g}")
    if v.get("FETCH_SIZE") or v.get("WRITE_SIZE"):
        rd, wr = 2 * v["FETCH_SIZE"] * 1024 / n, v["WRITE_SIZE"] * 1024 / n
        print(f"    HBM read {rd / 1e6:9.1f} MB  write {wr / 1e6:9.1f} MB  -> {(rd + wr) / (ms * 1e-3) / 1e12 if ms else 0:.2f} TB/s")
