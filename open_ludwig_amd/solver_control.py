"""Time-step control: host mirror of src/solver_control.jl (the caller of the drop-in boundary).

Same call order and buffer parity as the reference: per level, iseven(t_sub) picks (f, vel) as input and
(f_temp, vel_temp) as output, else swapped (:35-41); a level with children saves its input state before it steps
(:46-48) and then steps its child twice, at 2*t_sub with temporal weight 0.0 and at 2*t_sub+1 with 0.5 (:63-83).
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .blocks import DeviceLevel, has_temporal_storage
from .physics import SolverParams, perform_timestep_v2


def recursive_step_temporal(grids: Sequence[DeviceLevel], current_lvl: int, t_sub: int,
                            parent: Optional[DeviceLevel], parent_tau, temporal_weight, u_vel,
                            params: SolverParams) -> None:
    """recursive_step_temporal! (src/solver_control.jl:86-143). current_lvl is 1-based."""
    if current_lvl > len(grids):
        return
    level = grids[current_lvl - 1]
    has_children = current_lvl < len(grids)
    if has_children and params.use_temporal_interp and has_temporal_storage(level):
        level.copy_to_old(t_sub)                       # copy_to_old!(level, f_in, vel_in)
    perform_timestep_v2(level, parent, parent_tau, u_vel, params, t_sub, temporal_weight)
    if has_children:
        recursive_step_temporal(grids, current_lvl + 1, 2 * t_sub, level, level.tau, np.float32(0.0), u_vel, params)
        recursive_step_temporal(grids, current_lvl + 1, 2 * t_sub + 1, level, level.tau, np.float32(0.5), u_vel, params)


def recursive_step(grids: Sequence[DeviceLevel], current_lvl: int, t_sub: int, parent: Optional[DeviceLevel],
                   parent_tau, u_vel, params: SolverParams) -> None:
    """recursive_step! (src/solver_control.jl:21-84): the same body with the temporal weight fixed at 0.0f0."""
    recursive_step_temporal(grids, current_lvl, t_sub, parent, parent_tau, np.float32(0.0), u_vel, params)


def execute_timestep_batch(grids: Sequence[DeviceLevel], t_start: int, batch_size: int, u_curr,
                           params: SolverParams, native: bool = True) -> None:
    """execute_timestep_batch! (src/solver_control.jl:145-165); t_start is 1-based like the reference's loop.

    native=True (default): the whole batch is one C call (ludwig_execute_timestep_batch runs the same recursion inside
    the library, so a multi-level coarse step is not paced by Python). native=False: the recursion of this module, call
    by call - the two are tested to give identical results."""
    if native:
        import ctypes as C
        from . import _lib
        arr = (C.c_void_p * len(grids))(*[g.handle for g in grids])
        fl = params.to_c()
        _lib.check(_lib.load().ludwig_execute_timestep_batch(arr, len(grids), int(t_start), int(batch_size),
                                                             float(np.float32(u_curr)), C.byref(fl)))
        return
    for t_offset in range(batch_size):
        t = t_start + t_offset
        recursive_step(grids, 1, t, None, np.float32(0.5), u_curr, params)
    grids[0].synchronize()                             # KernelAbstractions.synchronize(backend)


def ramp_velocity(batch_end: int, ramp_steps: int, u_target) -> np.float32:
    """Inlet-speed ramp, evaluated once per batch at batch_end (src/main.jl:173-174), Float32 throughout."""
    if batch_end <= ramp_steps:
        arg = np.float32(np.pi) * np.float32(batch_end) / np.float32(ramp_steps)
        prog = np.float32(0.5) * (np.float32(1.0) - np.float32(np.cos(np.float64(arg))))
    else:
        prog = np.float32(1.0)
    return np.float32(u_target) * prog
