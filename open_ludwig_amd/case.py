"""Case driver: the time loop of src/main.jl:169-232 around the HIP engine (YAML + STL in, Cd/Cl series out).

Loop semantics kept from the reference: steps run in batches of
`gpu.async_depth`; the inlet speed is the cosine ramp evaluated ONCE per batch at `batch_end`; diagnostics fire when
`batch_end % diag_freq < batch_length` and look at the state at batch END (stats from level 1's `rho`, forces from the
finest level's `rho` and `vel` buffer).

The stepping backend is injected (`stepper`), so the same loop drives the HIP library (HipStepper, the product path)
and, in tests only, the CPU oracle.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from . import forces as forces_mod
from .blocks import adapt
from .preprocess import CaseConfig, DomainParameters, setup_multilevel_domain, solver_params
from .solver_control import execute_timestep_batch, ramp_velocity


@dataclass
class DiagRow:
    step: int
    u_lat: float
    rho_min: float
    cd: float
    cl: float
    cs: float = 0.0
    cmy: float = 0.0


class HipStepper:
    """grids on one MI355X behind libludwig_hip.so"""

    def __init__(self, host_grids, device: int = 0):
        self.host = host_grids
        self.dev = [adapt(g, device, upload_state=False) for g in host_grids]
        for d in self.dev:
            d.init_equilibrium()               # src/main.jl:126-135 (every state array: nothing of the host's to upload first)

    def batch(self, t_start: int, n: int, u_curr, params) -> None:
        execute_timestep_batch(self.dev, t_start, n, u_curr, params)

    def field(self, level: int, name: str) -> np.ndarray:
        return self.dev[level].download(name)

    def surface_stresses(self, level: int, mesh, params, search_radius: int = 5):
        """per-triangle p, tau on the device (src/forces/surface.jl:138-266); reads the level's `vel` buffer like the reference"""
        h = self.host[level]
        return forces_mod.map_surface_stresses_device(mesh, self.dev[level], h.dx, h.tau, params, search_radius, "vel")

    def rho_min(self, level: int) -> float:
        """compute_flow_stats (src/diagnostics.jl:56-94), reduced on the device"""
        return self.dev[level].rho_min()

    def close(self):
        for d in self.dev:
            d.close()


class DistributedStepper:
    """grids spread over the ranks of the default torch.distributed group, one MI355X per rank (scope row N3): every
    level is cut on its own into equal parts (partition.level_owners; pass `owners` to choose otherwise), halo +
    parent-data ghosts move after every level step (partition.MultiLevelRunner).

    Diagnostics follow SURVEY section 8e: nothing but scalars crosses ranks on a diagnostics step.
      * rho_min: every rank reduces its owned level-1 cells on its device, one all-reduce MIN.
      * forces: the fluid cell a triangle reads (map_stresses_kernel!, src/forces/surface.jl:138-266) depends on the geometry
        only, so it is found once, on the host, from the global obstacle mask every rank holds; a triangle belongs to the rank
        that owns that cell. Per diagnostics step a rank gathers rho, u of ITS cells on its device (4 floats per owned
        triangle), evaluates the stresses and the nine Float32 sums of integrate_forces_kernel! over its triangles, and one
        all-gather of 10 floats per rank follows; every rank adds the partial sums in rank order. The sums of a rank are
        pairwise Float32 sums over its triangles, so the total differs from the single-device row by Float32 rounding of a
        different summation order (observed <= 1e-6 relative) - bit-equality with one device is traded for not moving fields.
    `field()` still assembles a GLOBAL array, on rank 0 only; it is used on VTU output steps."""

    def __init__(self, host_grids, device: Optional[int] = None, owners=None, stage_through_host: bool = False,
                 overlap: Optional[bool] = None, transport: Optional[str] = None):
        import torch
        import torch.distributed as dist
        from . import partition
        self.dist, self.partition, self.torch = dist, partition, torch
        self.host = host_grids
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device if device is not None else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(self.device)
        self.owners = owners if owners is not None else partition.level_owners(host_grids, self.world)
        self.stage = stage_through_host
        # overlap: each level's exchange runs under the part of its blocks that reads no ghost (partition.MultiLevelRunner);
        # LUDWIG_NO_OVERLAP=1 (or overlap=False) falls back to step -> exchange -> wait, level by level. transport: "native" = RCCL
        # called from the library (default with the nccl backend), "torch" = torch.distributed from Python (the gloo rehearsals).
        self.overlap = (os.environ.get("LUDWIG_NO_OVERLAP") is None) if overlap is None else bool(overlap)
        self.transport = transport
        self.runner = None
        self._tri = {}                         # level -> static triangle map (see surface_forces)

    def _start(self, params) -> None:
        self.runner = self.partition.MultiLevelRunner(self.host, self.owners, params, self.rank, self.world, self.device, self.stage,
                                                      overlap=self.overlap, transport=self.transport, upload_state=False)
        for lv in self.runner.levels:
            if lv is not None:
                lv.init_equilibrium()          # src/main.jl:126-135 (ghost blocks included: same rest state everywhere)

    def batch(self, t_start: int, n: int, u_curr, params) -> None:
        if self.runner is None:
            self._start(params)
        self.runner.params = params
        for t in range(t_start, t_start + n):
            self.runner.step(t, u_curr)
        self.runner.synchronize()

    # -- collectives of a few scalars --
    def _comm_device(self):
        return self.torch.device("cuda", self.device) if self.dist.get_backend() == "nccl" else self.torch.device("cpu")

    def rho_min(self, level: int) -> float:
        lv = self.runner.levels[level]
        mine = lv.rho_min() if (lv is not None and self.runner.views[level].n_owned > 0) else float("inf")
        # a diverged rank reports NaN (the reference's minimum() propagates it, src/diagnostics.jl:71); what MIN makes of a NaN is
        # the backend's business, so it travels as a flag: [min of the finite values, -1 if any rank saw NaN], one all-reduce MIN
        nan = mine != mine
        t = self.torch.tensor([float("inf") if nan else mine, -1.0 if nan else 0.0], dtype=self.torch.float32, device=self._comm_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float("nan") if float(t[1].item()) < 0 else float(t[0].item())

    def _triangle_map(self, level: int, mesh, params, search_radius: int):
        key = (level, search_radius)
        if key not in self._tri:
            g, view = self.host[level], self.runner.views[level]
            nc = forces_mod.nearest_fluid_cells(mesh, g.obstacle, g.block_pointer, g.dx, params, search_radius)
            owner = np.asarray(self.owners[level] if not isinstance(self.owners, np.ndarray) else
                               self.partition.ancestor_owner(g.level_id, g.active_block_coords, self.host[0].active_block_coords, self.owners))
            sel = np.flatnonzero(nc.found & (owner[nc.block] == self.rank))
            g2l = np.full(g.n_blocks, -1, dtype=np.int64)
            g2l[view.local_to_global[: view.n_owned]] = np.arange(view.n_owned)
            lb = g2l[nc.block[sel]]
            assert (lb >= 0).all()
            cell = nc.lx[sel] + 8 * nc.ly[sel] + 64 * nc.lz[sel] + 512 * lb
            sk = 512 * view.level.n_blocks
            dev = self.torch.device("cuda", self.device)
            idx = {"rho": self.torch.as_tensor(cell, dtype=self.torch.int64, device=dev),
                   "vel": self.torch.as_tensor(np.concatenate([cell, cell + sk, cell + 2 * sk]), dtype=self.torch.int64, device=dev)}
            buf = {k: self.torch.empty(v.numel(), dtype=self.torch.float32, device=dev) for k, v in idx.items()}
            self._tri[key] = (nc, sel, idx, buf)
        return self._tri[key]

    def surface_forces(self, level: int, mesh, params, symmetric: bool, search_radius: int = 5, want_maps: bool = False):
        """compute_aerodynamics! (src/forces/surface.jl:592-600) without moving fields: see the class docstring."""
        import ctypes as C
        from . import _lib
        nc, sel, idx, buf = self._triangle_map(level, mesh, params, search_radius)
        lv, g = self.runner.levels[level], self.host[level]
        lib = _lib.load()
        stream = C.c_void_p(self.torch.cuda.current_stream(self.torch.device("cuda", self.device)).cuda_stream)
        if sel.size:
            for name in ("rho", "vel"):      # the level's `vel` buffer, like the reference (src/forces/surface.jl:412)
                _lib.check(lib.ludwig_halo_pack(lv.handle, _lib.FIELD_NAMES[name], C.c_void_p(idx[name].data_ptr()), idx[name].numel(),
                                                C.c_void_p(buf[name].data_ptr()), stream))
            rho_c = buf["rho"].cpu().numpy()
            u_c = buf["vel"].cpu().numpy().reshape(3, -1).T
        else:
            rho_c, u_c = np.zeros(0, np.float32), np.zeros((0, 3), np.float32)
        p, tx, ty, tz = forces_mod.stress_from_cells(rho_c, np.ascontiguousarray(u_c), nc.wall_dist[sel], np.ones(sel.size, bool),
                                                      mesh.normals[sel], g.tau, params)
        part = np.zeros(10, dtype=np.float32)
        part[:9] = forces_mod.partial_force_sums(mesh, p, tx, ty, tz, params, select=sel)
        part[9] = np.count_nonzero(np.abs(p) > 1e-10)
        total, cov = forces_mod.combine_partial_sums(part, self._comm_device())      # fixed order: rank 0, 1, ...
        fr = forces_mod.finish_forces(total, cov, params, symmetric)
        if want_maps:                          # output steps only: per-triangle loads to rank 0 for the surface VTU
            parts = [None] * self.world if self.rank == 0 else None
            self.dist.gather_object((sel, p, tx, ty, tz), parts, dst=0)
            if self.rank == 0:
                n = mesh.centers.shape[0]
                maps = [np.zeros(n, dtype=np.float32) for _ in range(4)]
                for s2, *arrs in parts:
                    for m, a in zip(maps, arrs):
                        m[s2] = a
                fr.maps = tuple(maps)
        return fr

    def field(self, level: int, name: str) -> Optional[np.ndarray]:
        """GLOBAL array of a field, assembled on rank 0 (None elsewhere): result files only, never per diagnostics step."""
        lv, view = self.runner.levels[level], self.runner.views[level]
        g = self.host[level]
        mine = (view.local_to_global[: view.n_owned], lv.download(name)[:, :, :, : view.n_owned] if lv is not None else None)
        parts = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(mine, parts, dst=0)
        if self.rank != 0:
            return None
        out = np.zeros(getattr(g, name).shape, dtype=getattr(g, name).dtype, order="F")
        for l2g, a in parts:
            if a is not None and l2g.size:
                out[:, :, :, l2g] = a
        return out

    def close(self):
        if self.runner is not None:
            self.runner.close()          # plans, communicator, levels; the views and plans stay readable (statistics)


def _aerodynamics(st, grids, mesh, params, symmetric: bool, rho_f=None, want_maps: bool = False):
    """compute_aerodynamics! (src/forces/surface.jl:592-600) on the finest level: stresses on the device when the stepper
    offers it, else from downloaded fields; integration on the host either way. A distributed stepper reduces per rank."""
    fin = len(grids) - 1
    if hasattr(st, "surface_forces"):
        return st.surface_forces(fin, mesh, params, symmetric, want_maps=want_maps)
    if hasattr(st, "surface_stresses"):
        p, tx, ty, tz = st.surface_stresses(fin, mesh, params)
        fr = forces_mod.integrate_surface_forces(mesh, p, tx, ty, tz, params, symmetric)
        fr.maps = (p, tx, ty, tz)
        return fr
    if rho_f is None:
        rho_f = st.field(fin, "rho")
    return forces_mod.compute_aerodynamics(mesh, grids[fin], rho_f, st.field(fin, "vel"), params, symmetric)


def flow_stats(rho: np.ndarray, obstacle: np.ndarray) -> float:
    """rho_min of compute_flow_stats (src/diagnostics.jl:56-94, CUDA branch): minimum over non-obstacle cells"""
    return float(rho[~obstacle].min())


def run_case(cfg: CaseConfig, stepper_factory: Callable = HipStepper, steps: Optional[int] = None, stl_path: Optional[str] = None,
             log: Optional[Callable[[str], None]] = None, setup=None, out_dir: Optional[str] = None, write_files: bool = True):
    """solve_main (src/main.jl:54-249). Returns (rows, setup_report, params).

    out_dir: when given, the reference's result files are written there (row N4): convergence.csv and forces.csv at every
    diagnostics step, flow_%06d.vtu (+ surface_%06d.vtu) every `output_freq` steps. Unlike the reference (main.jl:79) an
    existing directory is NOT emptied first. write_files=False on all ranks but one of a distributed run."""
    import time as _time
    from . import output as out_mod
    grids, mesh, params, report = setup if setup is not None else setup_multilevel_domain(cfg, stl_path)
    sp = solver_params(cfg, params)
    st = stepper_factory(grids)
    total_steps = steps if steps is not None else cfg.steps
    rows: List[DiagRow] = []
    batch = cfg.async_depth
    t = 1
    writing = out_dir is not None and write_files
    if writing:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "convergence.csv"), "w") as io:
            io.write(out_mod.CONVERGENCE_CSV_HEADER + "\n")
        if cfg.forces_enabled:
            out_mod.write_force_csv_header(os.path.join(out_dir, "forces.csv"))
    t0 = last_diag = _time.time()
    total_cells = sum(g.n_blocks * 512 for g in grids)
    fr = None
    try:
        while t <= total_steps:
            batch_end = min(t + batch - 1, total_steps)
            actual = batch_end - t + 1
            u_curr = ramp_velocity(batch_end, cfg.ramp_steps, cfg.u_lattice)
            st.batch(t, actual, u_curr, sp)
            if batch_end % cfg.diag_freq < actual or batch_end == total_steps:
                diag_step = (batch_end // cfg.diag_freq) * cfg.diag_freq
                if t <= diag_step <= batch_end:
                    rho1 = None
                    if hasattr(st, "rho_min"):
                        rho_min = st.rho_min(0)
                    else:
                        rho1 = st.field(0, "rho")
                        rho_min = flow_stats(rho1, grids[0].obstacle)
                    cd = cl = cs = cmy = float("nan")
                    if cfg.forces_enabled:
                        fr = _aerodynamics(st, grids, mesh, params, cfg.symmetric_analysis, rho1 if len(grids) == 1 else None)
                        cd, cl, cs, cmy = fr.Cd, fr.Cl, fr.Cs, fr.Cmy
                    rows.append(DiagRow(diag_step, float(u_curr), rho_min, cd, cl, cs, cmy))
                    now = _time.time()
                    mlups = (total_cells * cfg.diag_freq) / (max(now - last_diag, 1e-9) * 1e6)     # src/main.jl:189
                    last_diag = now
                    if writing:
                        time_phys = float(diag_step) * params.time_scale
                        if cfg.forces_enabled:
                            out_mod.append_force_csv(os.path.join(out_dir, "forces.csv"), diag_step, time_phys, fr, u_curr)
                        with open(os.path.join(out_dir, "convergence.csv"), "a") as io:
                            io.write(out_mod.convergence_csv_row(diag_step, now - t0, time_phys, u_curr, rho_min, mlups,
                                                                 cd if cfg.forces_enabled else None, cl if cfg.forces_enabled else None) + "\n")
                    if log:
                        log(f"{diag_step:8d} | {float(u_curr):.4f} | {rho_min:.4f} | {cd:8.4f} | {cl:8.4f}")
            if out_dir is not None and batch_end % cfg.output_freq < actual:                      # src/main.jl:213-231
                out_step = (batch_end // cfg.output_freq) * cfg.output_freq
                if t <= out_step <= batch_end:
                    fetched = {}

                    def fields(lvl, name):
                        if name == "obstacle":
                            return grids[lvl].obstacle
                        if (lvl, name) not in fetched:
                            fetched[(lvl, name)] = st.field(lvl, name)          # collective in a distributed run
                        return fetched[(lvl, name)]

                    mesh_arrays_needed = out_mod.select_export_blocks([g.active_block_coords for g in grids])
                    vel_name = "vel_temp" if out_step % 2 == 0 else "vel"
                    for lvl in sorted({l for l, _ in mesh_arrays_needed}):
                        fields(lvl, "rho"); fields(lvl, vel_name)
                    if cfg.forces_enabled and (fr is None or fr.maps is None or out_step != (out_step // cfg.diag_freq) * cfg.diag_freq):
                        fr = _aerodynamics(st, grids, mesh, params, cfg.symmetric_analysis, want_maps=True)
                    if writing:
                        out_mod.export_merged_mesh(out_step, grids, fields, out_dir, cfg.output_fields)
                        if cfg.forces_enabled:
                            out_mod.save_surface_vtk(os.path.join(out_dir, "surface_%06d" % out_step), mesh, *fr.maps)
            t = batch_end + 1
    finally:
        if hasattr(st, "close"):
            st.close()
    return rows, report, params
