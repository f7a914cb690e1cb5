"""ctypes binding of libhyperpri_hip.so (the C ABI declared in include/hyperpri_hip.h).

The argument types are read from the header itself, so the header stays the single source of
truth for the boundary.  There is NO fallback: if the library is missing or a call fails, a
RuntimeError is raised -- the product path never routes through PyTorch ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
DIAG = os.environ.get("HPRI_DIAG", "0") == "1"      # the diagnostics build (hyperpri_amd/build.py, include/hyperpri_hip_diag.h)
LIB_PATH = os.path.join(_HERE, "lib", "libhyperpri_hip_diag.so" if DIAG else "libhyperpri_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "hyperpri_hip.h")
DIAG_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "hyperpri_hip_diag.h")

LIB_F16_PATH = os.path.join(_HERE, "lib", "libhyperpri_hip_f16.so")      # the same sources with IEEE half as the 16-bit type (precision "f16")

_lock = threading.Lock()
_lib = None
_lib_f16 = None
_decls = None
_tls = threading.local()        # .kind: which library the calls of this thread go to ("f16" inside an f16-mode tape, see ``using``)


def _ctype(decl: str):
    d = decl.strip()
    if "*" in d or "hipStream_t" in d:
        return ctypes.c_void_p
    if "unsigned long long" in d:
        return ctypes.c_ulonglong
    if "long long" in d:
        return ctypes.c_longlong
    if "size_t" in d:
        return ctypes.c_size_t
    if re.search(r"\bfloat\b", d):
        return ctypes.c_float
    if re.search(r"\bint\b", d):
        return ctypes.c_int
    raise ValueError(f"unhandled C type in header: {decl!r}")


def parse_header(path: str = HEADER_PATH):
    """{name: (restype, [argtypes])} for every ``hpri_*`` function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(hpri_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.endswith("*"):
            restype = ctypes.c_char_p
        elif "size_t" in ret:
            restype = ctypes.c_size_t
        else:
            restype = ctypes.c_int
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        out[name] = (restype, argtypes)
    return out


def load():
    """Load (once) and return the ctypes library; raises if it has not been built."""
    global _lib, _decls
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"hyperpri_amd: HIP extension not built ({LIB_PATH} missing). Run "
                "`python -m hyperpri_amd.build` (needs hipcc, --offload-arch=gfx950). "
                "There is no CPU/PyTorch fallback for the hot path.")
        lib = ctypes.CDLL(LIB_PATH)
        _decls = parse_header()
        if DIAG:
            _decls.update(parse_header(DIAG_HEADER_PATH))
        for name, (restype, argtypes) in _decls.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def load_f16():
    """The half-precision build of the same library (hyperpri_amd/build.py: LIB_F16); raises if it has not been built."""
    global _lib_f16
    if _lib_f16 is not None:
        return _lib_f16
    load()
    with _lock:
        if _lib_f16 is not None:
            return _lib_f16
        if not os.path.exists(LIB_F16_PATH):
            raise RuntimeError(f"hyperpri_amd: precision 'f16' needs {LIB_F16_PATH} (python -m hyperpri_amd.build)")
        lib = ctypes.CDLL(LIB_F16_PATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib_f16 = lib
    return _lib_f16


def current():
    """The library the calling thread's launches go to."""
    return load_f16() if getattr(_tls, "kind", None) == "f16" else load()


def kind() -> str:
    return "f16" if getattr(_tls, "kind", None) == "f16" else "bf16"


class using:
    """``with using("f16"):`` -- the launches of this thread go to the half-precision library (an f16-mode tape, forward and backward)."""

    def __init__(self, kind):
        self.kind = "f16" if kind == "f16" else None

    def __enter__(self):
        self.prev = getattr(_tls, "kind", None)
        _tls.kind = self.kind
        return self

    def __exit__(self, *exc):
        _tls.kind = self.prev
        return False


def call(name: str, *args) -> None:
    """Call an int-returning launcher; raise RuntimeError with the library's message on failure."""
    lib = current()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.hpri_last_error()
        raise RuntimeError(f"{name} failed (code {rc}): {msg.decode() if msg else ''}")
