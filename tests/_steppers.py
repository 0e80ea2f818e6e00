"""Test-only stepping backends for open_ludwig_amd.case.run_case (the product path uses HipStepper)."""
import numpy as np

from oracle import oracle


class OracleStepper:
    """the CPU oracle behind the same interface as HipStepper; steps host BlockLevels in place"""

    def __init__(self, host_grids):
        self.grids = host_grids
        for g in host_grids:
            oracle.init_equilibrium(g)           # src/main.jl:126-135

    def batch(self, t_start, n, u_curr, params):
        oracle.execute_timestep_batch(self.grids, t_start, n, np.float32(u_curr), params)

    def field(self, level, name):
        return getattr(self.grids[level], name)

    def close(self):
        pass
