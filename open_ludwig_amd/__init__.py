"""open_ludwig_amd - MI355X-native D3Q27 collide-and-stream engine behind OPEN_Ludwig's per-level step API.

Layout
  csrc/             hand-written gfx950 HIP kernels + the C ABI (include/ludwig_hip.h) -> libludwig_hip.so
  _lib.py           ctypes binding of that library (no fallback: raises if the library is missing)
  blocks.py         BlockLevel / adapt / build_neighbor_table   (reference src/blocks.jl, src/domain_topology.jl)
  physics.py        perform_timestep_v2 / apply_bouzidi_correction (reference src/physics_v2.jl, src/bouzidi_kernel.jl)
  solver_control.py recursive_step / execute_timestep_batch     (reference src/solver_control.jl)
  partition.py      static block partition + one-cell halo exchange over torch.distributed (RCCL)
  cases.py          synthetic BlockLevel builders for tests and bench.py
"""
from .blocks import BLOCK_SIZE, BlockLevel, DeviceLevel, adapt, build_lattice_arrays, build_neighbor_table, has_temporal_storage
from .physics import SolverParams, apply_bouzidi_correction, perform_timestep_v2, stream_collide
from .solver_control import execute_timestep_batch, ramp_velocity, recursive_step, recursive_step_temporal

__all__ = [
    "BLOCK_SIZE", "BlockLevel", "DeviceLevel", "adapt", "build_lattice_arrays", "build_neighbor_table",
    "has_temporal_storage", "SolverParams", "apply_bouzidi_correction", "perform_timestep_v2", "stream_collide",
    "execute_timestep_batch", "ramp_velocity", "recursive_step", "recursive_step_temporal",
]
