"""Register-spill gate: no kernel of the hot path may spill a vector register or use scratch memory.

tools/check_codeobj.py compiles csrc/*.hip device-only for gfx950 (no GPU needed) and reads every kernel's register / spill /
scratch figures; the tolerated class (a spill proven to lie outside the k loop, bounded in size) is listed in the tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_hot_path_kernels_do_not_spill():
    import check_codeobj as C
    rep = C.run()
    names = {k["demangled"].split("(")[0].replace("void ", "") for k in rep["kernels"]}
    # the report really covers the kernels a step launches
    for must in ("conv_wino4_kernel", "conv_bf16v3_kernel<false>", "gemm_bf16v3_kernel<1>", "wino_wgrad_reduce_wide_kernel<16>",
                 "conv_wino_wgrad_kernel", "wgrad1x1_bf16v3_kernel<0>", "outconv_fwd_kernel<true, false>", "outconv_fwd_wide_kernel<true, true>"):
        assert must in names, f"{must} missing from the code-object report"
    hot = [k for k in rep["kernels"] if k["hot"]]
    assert len(hot) >= 40
    assert rep["ok"], "hot-path kernels with spills: " + "; ".join(rep["hot_path_kernels_with_spills"])
    for k in hot:
        if not k["clean"]:
            assert k["tolerated"] and k["scratch"] <= 16, k
