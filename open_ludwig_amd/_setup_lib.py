"""ctypes binding of libludwig_setup.so (include/ludwig_setup.h): the native host-side case set-up.

Built in-tree with g++ (`build()`); `load()` raises when it is missing - `preprocess.setup_multilevel_domain` uses it by default and
does not quietly fall back to the numpy restatement (that one is kept as the checker, `method="numpy"`)."""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCE = os.path.join(CSRC, "setup_host.cpp")
HEADER = os.path.join(_HERE, "..", "include", "ludwig_setup.h")
OUT = os.path.join(CSRC, "libludwig_setup.so")
# -ffp-contract=off: the reference's Float64 host code never fuses a*b+c; voxels and q values are compared bit for bit
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-pthread"]

EXPORTED_SYMBOLS = ["lws_last_error", "lws_voxelize", "lws_flood_fill", "lws_wall_distance", "lws_bouzidi_qmap", "lws_f64_to_f16"]

_lib = None


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in (SOURCE, HEADER))


def build(force: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise FileNotFoundError("g++ not found: cannot build libludwig_setup.so")
    res = subprocess.run([cxx, *CXX_FLAGS, SOURCE, "-o", OUT], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed building libludwig_setup.so:\n" + res.stderr)
    return OUT


def load():
    global _lib
    if _lib is not None:
        return _lib
    if needs_build():
        build()
    lib = C.CDLL(OUT)
    p = C.c_void_p
    lib.lws_last_error.restype = C.c_char_p
    lib.lws_voxelize.argtypes = [p, C.c_int64, C.c_double, p, C.c_int64, p, C.c_int]
    lib.lws_voxelize.restype = C.c_int
    lib.lws_flood_fill.argtypes = [p, C.c_int64, p]
    lib.lws_flood_fill.restype = C.c_int64
    lib.lws_wall_distance.argtypes = [p, C.c_int64, p, C.c_double, p, C.c_int]
    lib.lws_wall_distance.restype = C.c_int64
    lib.lws_bouzidi_qmap.argtypes = [p, C.c_int64, C.c_double, p, C.c_int64, p, p, C.c_int]
    lib.lws_bouzidi_qmap.restype = C.c_int64
    lib.lws_f64_to_f16.argtypes = [C.c_double]
    lib.lws_f64_to_f16.restype = C.c_uint16
    _lib = lib
    return lib


def check(ret: int, what: str) -> int:
    if ret < 0:
        raise RuntimeError(f"{what}: {load().lws_last_error().decode()}")
    return ret


def n_threads() -> int:
    """LUDWIG_SETUP_THREADS, else every hardware thread this process may use"""
    v = os.environ.get("LUDWIG_SETUP_THREADS")
    if v:
        return max(1, int(v))
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)
