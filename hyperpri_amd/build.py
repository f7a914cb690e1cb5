"""Build the HIP extension in-tree: hyperpri_amd/lib/libhyperpri_hip.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the
GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# HPRI_DIAG=1: the diagnostics build -- superseded kernel generations and neutral / negative variants kept for A/B measurements
# (include/hyperpri_hip_diag.h) compiled in with -DHPRI_DIAG_KERNELS, as a SEPARATE library; the product library has one kernel
# family per precision mode and form.
DIAG = os.environ.get("HPRI_DIAG", "0") == "1"
LIB = os.path.join(LIBDIR, "libhyperpri_hip_diag.so" if DIAG else "libhyperpri_hip.so")
# The product is TWO libraries built from the same sources: libhyperpri_hip.so (16-bit type of the plane paths = bf16: precision modes
# fp32 / bf16 / bf16x3 / bf16x6) and libhyperpri_hip_f16.so (-DHPRI_H16_F16: IEEE half, precision mode "f16"; csrc/common.h).
LIB_F16 = os.path.join(LIBDIR, "libhyperpri_hip_f16.so")
SOURCES = ["api.cpp", "conv_fwd.hip", *(["conv_bf16v2.hip"] if DIAG else []), "conv_bf16v3.hip", "gemm_bf16v3.hip", "gemm_f32v2.hip", "wgrad_bf16v3.hip", "conv_wino.hip", "conv_wino4.hip", "conv_wgrad.hip", "conv_wgrad_bf16v2.hip", "pack.hip", "bn.hip", "elementwise.hip", "step.hip", "ingest.hip", "conv_ingest.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", *(["-DHPRI_DIAG_KERNELS"] if DIAG else [])]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp(flags=None) -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if not os.path.isfile(os.path.join(CSRC, f)):
            continue
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    h.update(" ".join(FLAGS if flags is None else flags).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    """Build the library (both product libraries unless HPRI_DIAG=1); returns the path of libhyperpri_hip.so."""
    lib = _build_one(LIB, FLAGS, "_diag.o" if DIAG else ".o", force, verbose)
    if not DIAG:
        _build_one(LIB_F16, FLAGS + ["-DHPRI_H16_F16"], "_f16.o", force, verbose)
    return lib


def _build_one(LIB: str, FLAGS, osuffix: str, force: bool, verbose: bool) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_file = LIB + ".stamp"
    stamp = _stamp(FLAGS)
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, s.rsplit(".", 1)[0] + osuffix)
        objs.append(o)
        cmd = [hipcc, *FLAGS, "-x", "hip", "-c", os.path.join(CSRC, s), "-o", o, "-I", CSRC]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
