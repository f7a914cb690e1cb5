#!/usr/bin/env python3
"""Cycle shares of a conv_wino4 workgroup (4 waves, 16x8 pixels; two stamp groups = waves 0-1 / 2-3) (diagnostic build, tools/build_stamps.sh): main loop | output transform + stores |
statistics, median over workgroups."""
import os as _os
_os.environ.setdefault("HPRI_DIAG", "1")     # uses entry points of the DIAGNOSTICS build (include/hyperpri_hip_diag.h): HPRI_DIAG=1 python -m hyperpri_amd.build
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
SHAPES = [(2, 608, 968, 238, 64), (2, 304, 484, 128, 128)]
if os.environ.get("WINO_SHAPES"):      # e.g. "2,608,968,64,64;2,304,484,128,128"
    SHAPES = [tuple(int(v) for v in q.split(",")) for q in os.environ["WINO_SHAPES"].split(";")]
def rup(x, m): return (x + m - 1) // m * m
LIBNAME = os.environ.get("WINO_LIB", "libv2stamps.so")
print("library", LIBNAME)
lib = ctypes.CDLL(os.path.join(ROOT, "hyperpri_amd", "lib", LIBNAME))
lib.hpri_last_error.restype = ctypes.c_char_p
lib.hpri_wino_packed_floats.restype = ctypes.c_size_t
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (N, H, W, Cin, Cout) in SHAPES:
    cs, cout_pad = rup(Cin, 8), rup(Cout, 64)
    x = torch.zeros(N * H * W, cs, device=dev); x[:, :Cin] = torch.randn(N * H * W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.1
    b = torch.randn(Cout, device=dev)
    up = torch.empty(lib.hpri_wino_packed_floats(Cin, cout_pad), device=dev)
    assert lib.hpri_wino4_pack(P(w), P(up), ctypes.c_void_p(0), 0, Cin, Cout, cout_pad, Cin, st) == 0
    tl = ctypes.c_int(); lib.hpri_conv_wino4_plan(N, H, W, ctypes.byref(tl))
    stats = torch.zeros(tl.value * cout_pad * 4, device=dev)
    y = torch.zeros(N * H * W * Cout, device=dev)
    nwg = tl.value * (cout_pad // 64)
    stamps = torch.zeros(nwg * 2 * 8, dtype=torch.int64, device=dev)
    lib.hpri_wino4_set_stamps(P(stamps))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        if it == 2:
            e0.record()
        rc = lib.hpri_conv_wino4(P(x), cs, 0, P(up), P(b), P(y), Cout, 0, P(stats), N, H, W, cs, Cout, cout_pad, Cout, 0, st)
        assert rc == 0, lib.hpri_last_error()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = stamps.view(-1, 2, 8).cpu().double()
    names = ["main loop", "transform+store", "statistics"]
    print(f"N{N} {H}x{W} {Cin}->{Cout}: {nwg} workgroups, {Cin // 8 if Cin % 8 == 0 else cs // 8} stages")
    span = float(t[:, :, 3].max() - t[:, :, 0].min())
    print(f"   launch {ms:.3f} ms; first stamp -> last stamp {span:.0f} ticks = {span / ms / 1e6:.3f} GHz if s_memtime counts shader clocks")
    # co-residency: group workgroups by (XCC, SE, CU) from HW_ID [bits 8-11 CU, 13-15 SE on gfx9] and look at how the two
    # slots of a CU are phased: fraction of a workgroup's epilogue [t1, t3] that overlaps ANOTHER workgroup's main loop [t0, t1]
    raw = stamps.view(-1, 2, 8)[:, 0, :].cpu()
    hw = raw[:, 6] & 0xffffffff; xcc = (raw[:, 6] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    key = (xcc * 8 + se) * 32 + sh * 16 + cu
    import collections
    groups = collections.defaultdict(list)
    for i in range(raw.shape[0]):
        groups[int(key[i])].append((int(raw[i, 0]), int(raw[i, 1]), int(raw[i, 3]), i))
    print(f"   {len(groups)} distinct (xcc, se, sh, cu) keys; workgroups per key: min {min(len(v) for v in groups.values())} max {max(len(v) for v in groups.values())}")
    tot_epi = 0; tot_cov = 0
    for k, v in groups.items():
        for (s0, s1, s3, i) in v:
            epi = s3 - s1
            cov = 0
            for (o0, o1, o3, j) in v:
                if j == i: continue
                cov += max(0, min(s3, o1) - max(s1, o0))
            tot_epi += epi; tot_cov += min(cov, epi)
    print(f"   epilogue time covered by a co-resident workgroup's main loop: {100.0 * tot_cov / max(tot_epi, 1):.1f} %")
    # what a CU slot spends OUTSIDE a workgroup's stamped region: from one workgroup's last stamp to the first stamp of the next
    # workgroup that starts on the same CU (dispatch + kernel preamble), and the prologue (first stamp -> first operands landed)
    gaps = []
    for k, v in groups.items():
        starts = sorted(s0 for (s0, s1, s3, i) in v)
        import bisect
        for (s0, s1, s3, i) in v:
            j = bisect.bisect_right(starts, s3)
            if j < len(starts):
                gaps.append(starts[j] - s3)
    gaps.sort()
    pro = (raw[:, 7] - raw[:, 0]).double()
    if gaps:
        print(f"   end of a workgroup -> start of the next one on that CU: median {gaps[len(gaps)//2]} p10 {gaps[len(gaps)//10]} p90 {gaps[9*len(gaps)//10]} ticks;"
              f" prologue (start -> first operands landed): median {pro.median():.0f} p90 {pro.quantile(0.9):.0f}")
    k0 = sorted(groups)[0]
    print("   first key timeline (start, loop end, end, id):", [(a0 - groups[k0][0][0], a1 - groups[k0][0][0], a3 - groups[k0][0][0], i) for (a0, a1, a3, i) in sorted(groups[k0])[:6]])
    for g in (0, 1):
        d = [t[:, g, i + 1] - t[:, g, i] for i in range(3)]
        tot = t[:, g, 3] - t[:, g, 0]
        e = [t[:, g, 4] - t[:, g, 1], t[:, g, 5] - t[:, g, 4], t[:, g, 2] - t[:, g, 5]]
        print(f"   waves {2*g}-{2*g+1}: epilogue: barrier + column transform + exchange writes {e[0].median():.0f} | barrier {e[1].median():.0f} | reads + row transform + stores {e[2].median():.0f}")
        print(f"   waves {2*g}-{2*g+1}: total {tot.median():8.0f} | " + " | ".join(f"{n} {v.median():7.0f} ({100 * v.median() / tot.median():4.1f} %)" for n, v in zip(names, d)) + f" | per stage {d[0].median() / (cs // 8):6.0f}")
