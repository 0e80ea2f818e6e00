"""ctypes wrapper of the CPU oracle (oracle/ludwig_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
anything under open_ludwig_amd/. Pin status: see ludwig_oracle.h (anchored by the reference's run-log series).

It operates IN PLACE on the numpy arrays of host BlockLevel objects (open_ludwig_amd.blocks.BlockLevel), which use
the reference's memory layout.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LUDWIG_ORACLE_LIB") or os.path.join(_HERE, "libludwig_oracle.so")   # override: tests/oneoff_oracle_contraction.py only


class OracleLevel(C.Structure):
    _fields_ = [
        ("level_id", C.c_int32), ("n_blocks", C.c_int32), ("tau", C.c_float),
        ("grid_dim_x", C.c_int32), ("grid_dim_y", C.c_int32), ("grid_dim_z", C.c_int32),
        ("block_pointer", C.c_void_p), ("neighbor_table", C.c_void_p),
        ("map_x", C.c_void_p), ("map_y", C.c_void_p), ("map_z", C.c_void_p),
        ("rho", C.c_void_p), ("vel", C.c_void_p), ("vel_temp", C.c_void_p),
        ("f", C.c_void_p), ("f_temp", C.c_void_p), ("f_post_collision", C.c_void_p),
        ("f_old", C.c_void_p), ("rho_old", C.c_void_p), ("vel_old", C.c_void_p),
        ("has_temporal_storage", C.c_int32),
        ("obstacle", C.c_void_p), ("sponge", C.c_void_p), ("wall_dist", C.c_void_p),
        ("bouzidi_enabled", C.c_int32), ("n_boundary_cells", C.c_int32),
        ("bouzidi_q_map", C.c_void_p), ("bouzidi_cell_block", C.c_void_p),
        ("bouzidi_cell_x", C.c_void_p), ("bouzidi_cell_y", C.c_void_p), ("bouzidi_cell_z", C.c_void_p),
    ]


class OracleParams(C.Structure):
    _fields_ = [
        ("domain_nx", C.c_int32), ("domain_ny", C.c_int32), ("domain_nz", C.c_int32),
        ("is_symmetric", C.c_int32), ("wall_model_active", C.c_int32), ("use_temporal_interp", C.c_int32),
        ("sponge_blend_distributions", C.c_int32),
        ("c_wale", C.c_float), ("nu_sgs_background", C.c_float), ("inlet_turbulence", C.c_float),
        ("q_min_threshold", C.c_float),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    srcs = [os.path.join(_HERE, "ludwig_oracle.c"), os.path.join(_HERE, "ludwig_oracle.h"),
            os.path.join(_HERE, "..", "open_ludwig_amd", "csrc", "jl_math.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, f32, i32, i64 = C.c_void_p, C.c_float, C.c_int32, C.c_int64
        L.oracle_stream_collide.argtypes = [C.POINTER(OracleLevel), C.POINTER(OracleLevel), vp, vp, f32, vp, vp, vp, vp,
                                            f32, i64, f32, C.POINTER(OracleParams)]
        L.oracle_stream_collide.restype = None
        L.oracle_bouzidi_correction.argtypes = [C.POINTER(OracleLevel), vp, f32]
        L.oracle_bouzidi_correction.restype = None
        L.oracle_perform_timestep.argtypes = L.oracle_stream_collide.argtypes
        L.oracle_perform_timestep.restype = None
        L.oracle_execute_timestep_batch.argtypes = [C.POINTER(OracleLevel), i32, i64, i32, f32, C.POINTER(OracleParams)]
        L.oracle_execute_timestep_batch.restype = None
        L.oracle_init_equilibrium.argtypes = [C.POINTER(OracleLevel)]
        L.oracle_init_equilibrium.restype = None
        L.oracle_ramp_progress.argtypes = [i64, i64]
        L.oracle_ramp_progress.restype = f32
        L.oracle_gradient_noise.argtypes = [i32, i32, i32, i32]
        L.oracle_gradient_noise.restype = f32
        L.oracle_half_to_float.argtypes = [C.c_uint16]
        L.oracle_half_to_float.restype = f32
        L.oracle_lattice.argtypes = [vp] * 7
        L.oracle_lattice.restype = None
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _check(a: np.ndarray, dtype) -> int:
    if a.dtype != dtype and not (dtype == np.uint8 and a.dtype == np.bool_):
        raise TypeError(f"expected {dtype}, got {a.dtype}")
    if not (a.flags.f_contiguous or a.flags.c_contiguous and a.ndim <= 1):
        raise ValueError("array must be Fortran-contiguous (reference layout)")
    return a.ctypes.data


def to_c_level(level) -> OracleLevel:
    """View a host BlockLevel as the oracle's struct (no copies: the oracle updates the arrays in place)."""
    o = OracleLevel()
    o.level_id = level.level_id
    o.n_blocks = level.n_blocks
    o.tau = float(level.tau)
    o.grid_dim_x, o.grid_dim_y, o.grid_dim_z = level.grid_dim_x, level.grid_dim_y, level.grid_dim_z
    o.block_pointer = _check(level.block_pointer, np.int32) if level.block_pointer.size else None
    o.neighbor_table = _check(level.neighbor_table, np.int32)
    o.map_x, o.map_y, o.map_z = (_check(a, np.int32) for a in (level.map_x, level.map_y, level.map_z))
    for name in ("rho", "vel", "vel_temp", "f", "f_temp", "f_post_collision", "f_old", "rho_old", "vel_old", "sponge", "wall_dist"):
        setattr(o, name, _check(getattr(level, name), np.float32))
    o.has_temporal_storage = 1 if level.f_old.size > 27 else 0
    o.obstacle = _check(level.obstacle, np.uint8)
    o.bouzidi_enabled = 1 if level.bouzidi_enabled else 0
    o.n_boundary_cells = level.n_boundary_cells
    o.bouzidi_q_map = _check(level.bouzidi_q_map, np.float16)
    o.bouzidi_cell_block = _check(level.bouzidi_cell_block, np.int32) if level.bouzidi_cell_block.size else None
    o.bouzidi_cell_x = _check(level.bouzidi_cell_x, np.int8) if level.bouzidi_cell_x.size else None
    o.bouzidi_cell_y = _check(level.bouzidi_cell_y, np.int8) if level.bouzidi_cell_y.size else None
    o.bouzidi_cell_z = _check(level.bouzidi_cell_z, np.int8) if level.bouzidi_cell_z.size else None
    return o


def to_c_params(params) -> OracleParams:
    p = OracleParams()
    p.domain_nx, p.domain_ny, p.domain_nz = params.domain_nx, params.domain_ny, params.domain_nz
    p.is_symmetric = 1 if params.symmetric_analysis else 0
    p.wall_model_active = 1 if params.wall_model_active else 0
    p.use_temporal_interp = 1 if params.use_temporal_interp else 0
    p.sponge_blend_distributions = 1 if params.sponge_blend_dist else 0
    p.c_wale = float(np.float32(params.c_wale))
    p.nu_sgs_background = float(np.float32(params.nu_sgs_bg))
    p.inlet_turbulence = float(np.float32(params.inlet_turbulence))
    p.q_min_threshold = float(np.float32(params.q_min_threshold))
    return p


def execute_timestep_batch(grids: Sequence, t_start: int, batch_size: int, u_curr, params) -> None:
    """execute_timestep_batch! on host BlockLevels, in place."""
    arr = (OracleLevel * len(grids))(*[to_c_level(g) for g in grids])
    p = to_c_params(params)
    lib().oracle_execute_timestep_batch(arr, len(grids), int(t_start), int(batch_size), float(np.float32(u_curr)), C.byref(p))


def init_equilibrium(level) -> None:
    o = to_c_level(level)
    lib().oracle_init_equilibrium(C.byref(o))


def newest_buffers(level_index: int, t_last: int):
    """Names of the buffers holding the newest state after coarse step t_last (Appendix A.13):
    level 1: f_temp/vel_temp after even t, f/vel after odd t; level >= 2: always f/vel (last sub-step index is odd)."""
    if level_index == 0:
        return ("f_temp", "vel_temp") if t_last % 2 == 0 else ("f", "vel")
    return ("f", "vel")


def num_threads() -> int:
    return lib().oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(int(n))
