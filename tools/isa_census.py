#!/usr/bin/env python3
"""Static instruction census of a kernel of the HIP library (VERDICT r4 item 5 ii: where do the ~1.0 non-MFMA vector instructions per
MFMA of conv_bf16v3 come from?): the kernel's gfx950 assembly (hipcc -S) split at its first and last v_mfma -- "before" (set-up),
"loop" (the k loop: everything between the first and the last MFMA) and "after" (epilogue + next item's set-up) -- with each
region's instructions by class.  A static count weights every instruction once; the k loop's body runs Cin_pad / 64 times per item
and the rest once, which the "per_item" estimate applies.   usage: isa_census.py conv_bf16v3.hip conv_bf16v3_kernel [chunks_per_item]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, kern = sys.argv[1], sys.argv[2]
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
csrc = os.path.join(ROOT, "hyperpri_amd", "csrc")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", "-I", csrc,
                           os.path.join(csrc, src), "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
# the kernel's body: from its label to its s_endpgm
m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(kern), text, re.S | re.M)
name, body = m.group(1), m.group(2)
ins = [ln.strip().split()[0] for ln in body.splitlines() if ln.startswith("\t") and not ln.strip().startswith((";", "."))]
first = next(i for i, x in enumerate(ins) if x.startswith("v_mfma"))
last = len(ins) - 1 - next(i for i, x in enumerate(reversed(ins)) if x.startswith("v_mfma"))


def klass(x):
    if x.startswith("v_mfma"):
        return "mfma"
    if x.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "valu_lane_moves (SGPR spills / uniform values)"
    if x.startswith("v_"):
        return "valu_other"
    if x.startswith("ds_"):
        return "lds"
    if x.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if x.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_setprio")):
        return "sync"
    if x.startswith("s_"):
        return "salu"
    return "other"


def census(lo, hi):
    c = {}
    for x in ins[lo:hi]:
        c[klass(x)] = c.get(klass(x), 0) + 1
    return c


regions = {"before_first_mfma": census(0, first), "k_loop (first to last mfma, both copies of the unrolled body)": census(first, last + 1),
           "after_last_mfma": census(last + 1, len(ins))}
loop = regions["k_loop (first to last mfma, both copies of the unrolled body)"]
# the unrolled body holds two chunks: it runs chunks / 2 times per item
times = chunks / 2.0
per_item = {"mfma": loop.get("mfma", 0) * times}
for k in ("valu_other", "valu_lane_moves (SGPR spills / uniform values)"):
    per_item[k + " in the k loop"] = loop.get(k, 0) * times
    per_item[k + " outside"] = regions["before_first_mfma"].get(k, 0) * 0 + regions["after_last_mfma"].get(k, 0)
valu_in = per_item["valu_other in the k loop"] + per_item["valu_lane_moves (SGPR spills / uniform values) in the k loop"]
valu_out = per_item["valu_other outside"] + per_item["valu_lane_moves (SGPR spills / uniform values) outside"]
print(json.dumps({"kernel": name, "source": src, "static_instruction_counts": regions, "chunks_per_item": chunks,
                  "per_item_estimate": per_item, "non_mfma_valu_per_mfma": {"k_loop": round(valu_in / per_item["mfma"], 3),
                                                                            "epilogue_and_setup": round(valu_out / per_item["mfma"], 3)}}, indent=1))
