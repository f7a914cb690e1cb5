"""bench.py replays committed PMC figures (roofline.traffic, roofline_238to64.hbm_bytes_measured) only for the build of the library
they were recorded on: the files carry the library's source stamp, a mismatch yields null + the reason."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_stale_pmc_files_are_not_replayed(tmp_path, monkeypatch):
    import bench
    os.makedirs(tmp_path / "profiles")
    kern = {"conv_wino4_kernel(Wino4Args)": {"launches": 70, "read_bytes_per_launch": 4e8, "write_bytes_per_launch": 1e8, "hbm_bytes_per_launch": 5e8},
            "void conv_bf16v3_kernel<false>(ConvV3Args)": {"launches": 10, "read_bytes_per_launch": 8e8, "write_bytes_per_launch": 3e8,
                                                            "hbm_bytes_per_launch": 1.1e9}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "_lib_stamp", lambda: "stamp-of-the-loaded-library")
    for name in ("r09_pmc_traffic.json", "r09_first_conv_pmc_traffic.json"):
        json.dump({"library_stamp": "some-other-build", "kernels": kern}, open(tmp_path / "profiles" / name, "w"))
    t, src = bench.pmc_traffic("conv_winograd_f32<3,F(2x2)>")
    assert t is None and src.startswith("NOT REPLAYED") and "another build" in src
    t, src = bench.first_conv_traffic()
    assert t is None and src.startswith("NOT REPLAYED")
    # files without a stamp (rounds 1-3) are stale by definition
    json.dump({"kernels": kern}, open(tmp_path / "profiles" / "r09_pmc_traffic.json", "w"))
    assert bench.pmc_traffic("conv_winograd_f32<3,F(2x2)>")[0] is None
    # the same figures recorded on THIS build are replayed
    for name in ("r09_pmc_traffic.json", "r09_first_conv_pmc_traffic.json"):
        json.dump({"library_stamp": "stamp-of-the-loaded-library", "kernels": kern}, open(tmp_path / "profiles" / name, "w"))
    assert bench.pmc_traffic("conv_winograd_f32<3,F(2x2)>") == (500000000, os.path.join("profiles", "r09_pmc_traffic.json"))
    assert bench.first_conv_traffic()[0] == 1100000000
