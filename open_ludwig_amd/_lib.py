"""ctypes binding of libludwig_hip.so (include/ludwig_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, this module raises.
The product path never routes through oracle/ or any CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LUDWIG_HIP_LIB") or os.path.join(_HERE, "csrc", "libludwig_hip.so")   # override: diagnostics only

# enum LudwigField
F, F_TEMP, F_POST, F_OLD, RHO, RHO_OLD, VEL, VEL_TEMP, VEL_OLD, OBSTACLE, SPONGE, WALL_DIST = range(12)
FIELD_NAMES = {
    "f": F, "f_temp": F_TEMP, "f_post_collision": F_POST, "f_old": F_OLD, "rho": RHO, "rho_old": RHO_OLD,
    "vel": VEL, "vel_temp": VEL_TEMP, "vel_old": VEL_OLD, "obstacle": OBSTACLE, "sponge": SPONGE,
    "wall_dist": WALL_DIST,
}
# enum LudwigPart
PART_ALL, PART_BOUNDARY, PART_INTERIOR = 0, 1, 2

# every symbol include/ludwig_hip.h declares (tests check the .so exports exactly these)
EXPORTED_SYMBOLS = [
    "ludwig_abi_version", "ludwig_last_error", "ludwig_device_count",
    "ludwig_level_create", "ludwig_level_destroy", "ludwig_level_set_stream", "ludwig_level_add_post_collision_readers", "ludwig_level_set_order",
    "ludwig_level_upload", "ludwig_level_download", "ludwig_level_field_ptr",
    "ludwig_init_equilibrium", "ludwig_step", "ludwig_stream_collide", "ludwig_bouzidi_correction",
    "ludwig_save_old", "ludwig_execute_timestep_batch", "ludwig_sync", "ludwig_halo_pack", "ludwig_halo_unpack", "ludwig_level_info",
    "ludwig_map_surface_stresses", "ludwig_level_rho_min", "ludwig_level_block_order",
    "ludwig_stream_create", "ludwig_stream_destroy", "ludwig_level_field_layout", "ludwig_level_set_rho_store",
    "ludwig_comm_unique_id", "ludwig_comm_create", "ludwig_comm_destroy", "ludwig_comm_allreduce_f32",
    "ludwig_halo_plan_create", "ludwig_halo_plan_destroy", "ludwig_halo_exchange", "ludwig_halo_wait",
    "ludwig_halo_plan_pack", "ludwig_halo_plan_unpack", "ludwig_halo_plan_buffers", "ludwig_halo_plan_timing", "ludwig_halo_plan_in_stream",
    "ludwig_halo_plan_exchange_ms", "ludwig_step_distributed",
]
UNIQUE_ID_BYTES = 128
HALO_GROUPS = ("f", "vel", "f_post", "rho")      # group index of partition.FIELD_GROUPS in a LudwigHaloPlanDesc


class LudwigError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libludwig_hip error {code}: {msg}")
        self.code = code


class LevelHost(C.Structure):
    _fields_ = [
        ("level_id", C.c_int32), ("n_blocks", C.c_int32), ("n_owned", C.c_int32), ("tau", C.c_float),
        ("grid_dim_x", C.c_int32), ("grid_dim_y", C.c_int32), ("grid_dim_z", C.c_int32),
        ("block_pointer", C.c_void_p), ("neighbor_table", C.c_void_p),
        ("map_x", C.c_void_p), ("map_y", C.c_void_p), ("map_z", C.c_void_p),
        ("obstacle", C.c_void_p), ("sponge", C.c_void_p), ("wall_dist", C.c_void_p),
        ("enable_temporal_interpolation", C.c_int32), ("n_boundary_cells", C.c_int32),
        ("bouzidi_q_map", C.c_void_p), ("bouzidi_cell_block", C.c_void_p),
        ("bouzidi_cell_x", C.c_void_p), ("bouzidi_cell_y", C.c_void_p), ("bouzidi_cell_z", C.c_void_p),
        ("comm_boundary", C.c_void_p), ("store_post_collision_everywhere", C.c_int32),
    ]


class StepFlags(C.Structure):
    _fields_ = [
        ("domain_nx", C.c_int32), ("domain_ny", C.c_int32), ("domain_nz", C.c_int32),
        ("is_symmetric", C.c_int32), ("wall_model_active", C.c_int32), ("use_temporal_interp", C.c_int32),
        ("sponge_blend_distributions", C.c_int32),
        ("c_wale", C.c_float), ("nu_sgs_background", C.c_float), ("inlet_turbulence", C.c_float),
        ("q_min_threshold", C.c_float),
    ]


class SurfaceParams(C.Structure):
    _fields_ = [
        ("dx", C.c_float), ("tau", C.c_float), ("offset_x", C.c_float), ("offset_y", C.c_float), ("offset_z", C.c_float),
        ("pressure_scale", C.c_float), ("stress_scale", C.c_float), ("search_radius", C.c_int32),
    ]


class HaloPlanDesc(C.Structure):
    _fields_ = [
        ("n_peers", C.c_int32), ("peer_ranks", C.c_void_p),
        ("send_count", C.c_void_p * 4), ("recv_count", C.c_void_p * 4),
        ("send_index", C.c_void_p * 4), ("recv_index", C.c_void_p * 4),
    ]


class LevelInfo(C.Structure):
    _fields_ = [
        ("n_blocks", C.c_int32), ("n_owned", C.c_int32), ("n_fast_blocks", C.c_int32),
        ("n_general_blocks", C.c_int32), ("n_boundary_cells", C.c_int32),
        ("has_temporal_storage", C.c_int32), ("has_post_collision", C.c_int32), ("n_xrun_blocks", C.c_int32),
        ("device_bytes", C.c_int64),
    ]


_lib = None


def load() -> C.CDLL:
    """Load libludwig_hip.so; raises if it has not been built (run `python __graft_entry__.py` / build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    sig = {
        "ludwig_abi_version": (C.c_int, []),
        "ludwig_last_error": (C.c_char_p, []),
        "ludwig_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "ludwig_level_create": (C.c_int, [C.POINTER(LevelHost), i32, C.POINTER(vp)]),
        "ludwig_level_destroy": (None, [vp]),
        "ludwig_level_set_stream": (C.c_int, [vp, vp]),
        "ludwig_level_add_post_collision_readers": (C.c_int, [vp, vp, i64]),
        "ludwig_level_set_order": (C.c_int, [vp, i32, vp, i64]),
        "ludwig_level_upload": (C.c_int, [vp, i32, vp, C.c_size_t]),
        "ludwig_level_download": (C.c_int, [vp, i32, vp, C.c_size_t]),
        "ludwig_level_field_ptr": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "ludwig_init_equilibrium": (C.c_int, [vp]),
        "ludwig_step": (C.c_int, [vp, vp, i64, f32, f32, f32, C.POINTER(StepFlags)]),
        "ludwig_stream_collide": (C.c_int, [vp, vp, i64, f32, f32, f32, C.POINTER(StepFlags), i32]),
        "ludwig_bouzidi_correction": (C.c_int, [vp, i64, f32]),
        "ludwig_save_old": (C.c_int, [vp, i64]),
        "ludwig_execute_timestep_batch": (C.c_int, [C.POINTER(vp), i32, i64, i32, f32, C.POINTER(StepFlags)]),
        "ludwig_sync": (C.c_int, [vp]),
        "ludwig_halo_pack": (C.c_int, [vp, i32, vp, i64, vp, vp]),
        "ludwig_halo_unpack": (C.c_int, [vp, i32, vp, i64, vp, vp]),
        "ludwig_map_surface_stresses": (C.c_int, [vp, i32, i32, vp, vp, C.POINTER(SurfaceParams), vp, vp, vp, vp]),
        "ludwig_level_info": (C.c_int, [vp, C.POINTER(LevelInfo)]),
        "ludwig_level_rho_min": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "ludwig_level_block_order": (C.c_int, [vp, vp]),
        "ludwig_stream_create": (C.c_int, [i32, i32, C.POINTER(vp)]),
        "ludwig_stream_destroy": (C.c_int, [i32, vp]),
        "ludwig_level_field_layout": (C.c_int, [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "ludwig_level_set_rho_store": (C.c_int, [vp, i32]),
        "ludwig_comm_unique_id": (C.c_int, [vp]),
        "ludwig_comm_create": (C.c_int, [vp, i32, i32, i32, C.POINTER(vp)]),
        "ludwig_comm_destroy": (None, [vp]),
        "ludwig_comm_allreduce_f32": (C.c_int, [vp, vp, i32, i32]),
        "ludwig_halo_plan_create": (C.c_int, [vp, vp, C.POINTER(HaloPlanDesc), C.POINTER(vp)]),
        "ludwig_halo_plan_destroy": (None, [vp]),
        "ludwig_halo_exchange": (C.c_int, [vp, i32, vp, vp]),
        "ludwig_halo_wait": (C.c_int, [vp]),
        "ludwig_halo_plan_pack": (C.c_int, [vp, i32, i32, vp]),
        "ludwig_halo_plan_unpack": (C.c_int, [vp, i32, i32, vp]),
        "ludwig_halo_plan_buffers": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64)]),
        "ludwig_halo_plan_timing": (C.c_int, [vp, i32]),
        "ludwig_halo_plan_in_stream": (C.c_int, [vp, i32]),
        "ludwig_halo_plan_exchange_ms": (C.c_int, [vp, vp, i32, C.POINTER(C.c_int32)]),
        "ludwig_step_distributed": (C.c_int, [vp, vp, vp, i64, f32, f32, f32, C.POINTER(StepFlags)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    if lib.ludwig_abi_version() != 1:
        raise RuntimeError("libludwig_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().ludwig_last_error()
        raise LudwigError(rc, msg.decode("utf-8", "replace") if msg else "")


def device_count() -> int:
    n = C.c_int(0)
    rc = load().ludwig_device_count(C.byref(n))
    return n.value if rc == 0 else 0
