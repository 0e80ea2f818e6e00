"""The C-ABI shared library: loads without a GPU, exports every symbol include/ludwig_hip.h declares (and nothing the
header does not declare), reports errors through codes + ludwig_last_error, and never needs oracle/."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from open_ludwig_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "ludwig_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ludwig_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_loads_without_gpu():
    build.build_library()
    lib = _lib.load()
    assert lib.ludwig_abi_version() == 1


def test_exports_match_header_exactly():
    build.build_library()
    declared = _header_functions()
    assert declared == sorted(_lib.EXPORTED_SYMBOLS), "binding list out of sync with include/ludwig_hip.h"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\bT (ludwig_[a-z_0-9]+)", out)))
    assert exported == declared
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None


def test_header_cites_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "ludwig_hip.h")).read()
    for cite in ("src/physics_v2.jl:26-97", "src/bouzidi_kernel.jl:99-123", "src/blocks.jl:199-205", "src/blocks.jl:67-87",
                 "src/solver_control.jl:35-41", "src/main.jl:109-134", "src/solver_control.jl:164"):
        assert cite in text, cite


def test_error_reporting_without_device():
    """Argument errors come back as codes with a message; with no GPU, create() must fail loudly, not fall back."""
    lib = _lib.load()
    n = C.c_int(-1)
    lib.ludwig_device_count(C.byref(n))
    assert n.value >= 0
    out = C.c_void_p()
    assert lib.ludwig_level_create(None, 0, C.byref(out)) == -1          # LUDWIG_ERR_INVALID
    assert b"null" in lib.ludwig_last_error()
    assert lib.ludwig_sync(None) == -1
    assert lib.ludwig_step(None, None, 1, 0.0, 0.5, 0.0, None) == -1
    if n.value == 0:
        from open_ludwig_amd import cases
        from open_ludwig_amd.blocks import adapt
        grids, _ = cases.periodic_box((1, 1, 1))
        with pytest.raises(_lib.LudwigError) as e:
            adapt(grids[0], 0)
        assert e.value.code == -3                                         # LUDWIG_ERR_NO_DEVICE
        st = C.c_void_p(1)
        assert lib.ludwig_stream_create(0, 8, C.byref(st)) == -3 and st.value is None     # no device: no stream, no fallback
    assert lib.ludwig_stream_create(0, 8, None) == -1
    assert lib.ludwig_stream_destroy(0, None) == 0                        # destroying "no stream" is a no-op


def test_product_package_never_imports_the_oracle():
    """No file of the product package - and none of the measurement helpers under tools/, the Julia binding or the headers - imports,
    loads or links anything under oracle/ (comments may mention it). Checkers that do live under tests/ (test_*.py, oneoff_*.py)."""
    bad = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)|libludwig_oracle|ludwig_oracle\.h|oracle[/\\]", re.M)
    for top in ("open_ludwig_amd", "tools", "julia", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".sh", ".jl")):
                    src = open(os.path.join(dirpath, f)).read()
                    code = "\n".join(l for l in src.splitlines() if not l.strip().startswith(("#", "//", "*", "/*")))
                    code = re.sub(r'\"\"\".*?\"\"\"', "", code, flags=re.S)
                    m = bad.search(code)
                    assert m is None, (os.path.join(dirpath, f), m.group(0))


# ---- the header as a C99 translation unit: struct layouts against the ctypes and Julia mirrors, calls through dlopen ----

_JULIA_TYPES = {"Int32": (4, 4), "Float32": (4, 4), "Int64": (8, 8), "UInt8": (1, 1), "Int8": (1, 1)}   # name -> (size, align)


def _julia_struct_layout(text, name):
    """C layout of an immutable Julia struct of plain fields (Julia lays isbits structs out like C): [(field, offset, size)]"""
    body = re.search(r"^struct\s+" + name + r"\b(.*?)^end", text, flags=re.S | re.M).group(1)
    body = re.sub(r"#.*", "", body)
    fields, off, max_al = [], 0, 1
    for fname, ftype in re.findall(r"(\w+)::((?:NTuple\{\d+,\s*)?[\w{}]+)", body):
        m = re.match(r"NTuple\{(\d+),\s*(.*)", ftype)                 # NTuple{N,T}: N elements of T inline, like a C array
        count, ftype = (int(m.group(1)), m.group(2).rstrip("}")) if m else (1, ftype)
        size, al = (8, 8) if ftype.startswith("Ptr{") else _JULIA_TYPES[ftype]
        size *= count
        off = (off + al - 1) // al * al
        fields.append((fname, off, size))
        off += size
        max_al = max(max_al, al)
    return fields, (off + max_al - 1) // max_al * max_al


@pytest.fixture(scope="module")
def abi_report(tmp_path_factory):
    build.build_library()
    exe = str(tmp_path_factory.mktemp("abi") / "abi_check")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "abi_check.c"), "-o", exe, "-ldl"], check=True)
    out = subprocess.run([exe, _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    rep = {"struct": {}, "field": {}, "enum": {}, "call": {}}
    for line in out.splitlines():
        kind, *rest = line.split(" ", 2)
        if kind == "struct":
            rep["struct"][rest[0]] = int(rest[1])
        elif kind == "field":
            sname, fname, o, sz = line.split()[1:]
            rep["field"].setdefault(sname, []).append((fname, int(o), int(sz)))
        elif kind == "enum":
            rep["enum"][rest[0]] = int(rest[1])
        elif kind == "call":
            rep["call"][rest[0]] = rest[1] if len(rest) > 1 else ""
    return rep


def test_header_is_c99_and_struct_layouts_match_ctypes(abi_report):
    pairs = {"LudwigLevelHost": _lib.LevelHost, "LudwigStepFlags": _lib.StepFlags, "LudwigSurfaceParams": _lib.SurfaceParams,
             "LudwigLevelInfo": _lib.LevelInfo, "LudwigHaloPlanDesc": _lib.HaloPlanDesc}
    for cname, cls in pairs.items():
        assert abi_report["struct"][cname] == C.sizeof(cls), cname
        want = [(n, getattr(cls, n).offset, getattr(cls, n).size) for n, _ in cls._fields_]
        assert abi_report["field"][cname] == want, cname          # same names, same order, same offsets and sizes
    assert abi_report["enum"] == {"LUDWIG_FIELD_COUNT": len(_lib.FIELD_NAMES), "LUDWIG_WALL_DIST": _lib.WALL_DIST,
                                  "LUDWIG_PART_INTERIOR": _lib.PART_INTERIOR, "LUDWIG_HALO_GROUPS": len(_lib.HALO_GROUPS),
                                  "LUDWIG_UNIQUE_ID_BYTES": _lib.UNIQUE_ID_BYTES}


def test_julia_binding_structs_match_the_header(abi_report):
    """julia/LudwigHIP.jl cannot run here (no Julia in the image); its struct definitions are at least layout-checked
    against what gcc makes of the header, and every symbol it ccalls must be one the header declares."""
    text = open(os.path.join(ROOT, "julia", "LudwigHIP.jl")).read()
    for jname, cname in (("LevelHost", "LudwigLevelHost"), ("StepFlags", "LudwigStepFlags"), ("HaloPlanDesc", "LudwigHaloPlanDesc")):
        fields, size = _julia_struct_layout(text, jname)
        assert fields == abi_report["field"][cname], jname
        assert size == abi_report["struct"][cname], jname
    called = set(re.findall(r"ccall\(\(:(\w+), LIB\)", text))
    assert called and called <= set(_header_functions()), called - set(_header_functions())
    enum_line = re.search(r"const F, F_TEMP.*?= Int32\.\(0:(\d+)\)", text)
    assert int(enum_line.group(1)) + 1 == abi_report["enum"]["LUDWIG_FIELD_COUNT"]


def test_c_caller_through_dlopen(abi_report):
    c = abi_report["call"]
    assert c["abi_version"] == "1 1"
    assert c["level_create_null"] == "-1 out_cleared"
    assert "null" in c["last_error"]
    assert c["level_create_bad"] == "-1" and c["level_info_null"] == "-1"


def test_multi_gpu_entry_points_reject_bad_arguments_without_a_device():
    """The communicator / halo-plan entry points (include/ludwig_hip.h "multi-GPU") answer argument errors with a code and a message,
    touch no GPU and load no RCCL doing so."""
    lib = _lib.load()
    out = C.c_void_p(1)
    assert lib.ludwig_comm_unique_id(None) == -1 and b"null" in lib.ludwig_last_error()
    assert lib.ludwig_comm_create(None, 0, 1, 0, C.byref(out)) == -1 and out.value is None          # the out pointer is cleared first
    ident = (C.c_char * _lib.UNIQUE_ID_BYTES)()
    assert lib.ludwig_comm_create(ident, 2, 2, 0, C.byref(out)) == -1                               # rank outside the world
    d = _lib.HaloPlanDesc()
    assert lib.ludwig_halo_plan_create(None, None, C.byref(d), C.byref(out)) == -1 and out.value is None
    assert lib.ludwig_halo_exchange(None, 0, None, None) == -1
    assert lib.ludwig_halo_wait(None) == -1
    assert lib.ludwig_step_distributed(None, None, None, 1, 0.0, 0.5, 0.0, None) == -1
    assert lib.ludwig_comm_allreduce_f32(None, None, 1, 0) == -1
    assert lib.ludwig_level_field_layout(None, 0, None, None, None) == -1
    lib.ludwig_comm_destroy(None)
    lib.ludwig_halo_plan_destroy(None)                                                               # destroying nothing is a no-op
