"""Per-level step API: host mirror of src/physics_v2.jl (the drop-in boundary).

perform_timestep_v2! (src/physics_v2.jl:26-97) launches the stream-collide kernel and then the Bouzidi
correction. Here both are HIP kernels behind libludwig_hip.so. The reference passes the A/B arrays explicitly
(f_out, f_in, vel_out, vel_in); its only callers (src/solver_control.jl:35-41) choose them from the parity of the
sub-step index, so this mirror takes that index and the library applies the same rule.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .blocks import DeviceLevel

KAPPA = np.float32(0.41)                             # src/physics_v2.jl:15
CS2_PHYSICS = np.float32(1.0) / np.float32(3.0)      # src/physics_v2.jl:16
CS4_PHYSICS = CS2_PHYSICS * CS2_PHYSICS              # src/physics_v2.jl:17


@dataclass
class SolverParams:
    """The scalar arguments execute_timestep_batch! forwards to every perform_timestep_v2! call
    (src/solver_control.jl:145-161) plus the two globals the step reads (SYMMETRIC_ANALYSIS, Q_MIN_THRESHOLD)."""
    domain_nx: int
    domain_ny: int
    domain_nz: int
    wall_model_active: bool = False
    c_wale: float = 0.5
    nu_sgs_bg: float = 0.0005
    inlet_turbulence: float = 0.0
    use_temporal_interp: bool = True
    sponge_blend_dist: bool = False
    symmetric_analysis: bool = False
    q_min_threshold: float = 0.001

    def to_c(self) -> _lib.StepFlags:
        fl = _lib.StepFlags()
        fl.domain_nx, fl.domain_ny, fl.domain_nz = int(self.domain_nx), int(self.domain_ny), int(self.domain_nz)
        fl.is_symmetric = 1 if self.symmetric_analysis else 0
        fl.wall_model_active = 1 if self.wall_model_active else 0
        fl.use_temporal_interp = 1 if self.use_temporal_interp else 0
        fl.sponge_blend_distributions = 1 if self.sponge_blend_dist else 0
        fl.c_wale = float(np.float32(self.c_wale))
        fl.nu_sgs_background = float(np.float32(self.nu_sgs_bg))
        fl.inlet_turbulence = float(np.float32(self.inlet_turbulence))
        fl.q_min_threshold = float(np.float32(self.q_min_threshold))
        return fl


def perform_timestep_v2(level: DeviceLevel, parent: Optional[DeviceLevel], parent_tau, u_curr, params: SolverParams,
                        timestep: int, temporal_weight=0.0) -> None:
    """perform_timestep_v2! (src/physics_v2.jl:26-97). `parent is None` <=> `parent_f === nothing` (level 1).

    f_in/f_out/vel_in/vel_out are selected from `timestep` parity exactly as src/solver_control.jl:35-41 does;
    the parent's newest state is the output of the parent's step `timestep >> 1` (src/solver_control.jl:63-83).
    Asynchronous: queued on the level's HIP stream (the reference synchronizes after each launch, F5).
    """
    if level.n_blocks == 0:
        return
    fl = params.to_c()
    lib = _lib.load()
    _lib.check(lib.ludwig_step(level.handle, parent.handle if parent is not None else None, int(timestep),
                               float(np.float32(u_curr)), float(np.float32(parent_tau)),
                               float(np.float32(temporal_weight)), C.byref(fl)))


def stream_collide(level: DeviceLevel, parent: Optional[DeviceLevel], parent_tau, u_curr, params: SolverParams,
                   timestep: int, temporal_weight=0.0, part: int = _lib.PART_ALL) -> None:
    """The stream_collide_kernel_v2! launch alone (src/physics_v2.jl:58-83), optionally on a subset of blocks."""
    fl = params.to_c()
    _lib.check(_lib.load().ludwig_stream_collide(
        level.handle, parent.handle if parent is not None else None, int(timestep), float(np.float32(u_curr)),
        float(np.float32(parent_tau)), float(np.float32(temporal_weight)), C.byref(fl), int(part)))


def apply_bouzidi_correction(level: DeviceLevel, timestep: int, q_min_threshold) -> None:
    """apply_bouzidi_correction! (src/bouzidi_kernel.jl:99-123) on the output buffer of step `timestep`."""
    _lib.check(_lib.load().ludwig_bouzidi_correction(level.handle, int(timestep), float(np.float32(q_min_threshold))))
