"""Synthetic case builders: BlockLevel inputs for tests, smoke() and bench.py.

The reference builds its BlockLevels from YAML + STL through host pre-processing (src/domain.jl:20-266), which is
scope row N1 and not built yet. These builders produce inputs of the same shape
analytically: full or partial block grids, a voxelised sphere with an analytic Bouzidi q-map, sponge and wall-distance
fields, nested 2:1 levels. They are host-side numpy code and contain no stepping logic.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .blocks import BLOCK_SIZE, BlockLevel, build_lattice_arrays, build_neighbor_table
from .physics import SolverParams

_CX, _CY, _CZ, _W, _OPP, _MY, _MZ = build_lattice_arrays()


def full_box_coords(nbx: int, nby: int, nbz: int) -> List[Tuple[int, int, int]]:
    """All blocks of an nbx x nby x nbz grid in the reference's order: sort of (bx,by,bz) tuples (src/domain.jl:171),
    i.e. bx slowest, bz fastest."""
    return [(bx, by, bz) for bx in range(1, nbx + 1) for by in range(1, nby + 1) for bz in range(1, nbz + 1)]


def make_level(level_id: int, coords: Sequence[Tuple[int, int, int]], dims: Tuple[int, int, int], tau: float,
               periodic=(False, False, False), temporal: bool = True, **kw) -> BlockLevel:
    coords = sorted(coords)
    table = build_neighbor_table(coords, dims[0], dims[1], dims[2], periodic)
    scale = 2 ** (level_id - 1)
    return BlockLevel(level_id, coords, table, 1.0 / scale, 1.0 / scale, tau, enable_temporal_interpolation=temporal, **kw)


def global_cell_coords(level: BlockLevel):
    """1-based global cell coordinates gx, gy, gz, each of shape (8,8,8,n_blocks) (src/physics_kernels.jl:46-48)."""
    B = BLOCK_SIZE
    loc = np.arange(1, B + 1)
    gx = (level.map_x.astype(np.int64)[None, None, None, :] - 1) * B + loc[:, None, None, None]
    gy = (level.map_y.astype(np.int64)[None, None, None, :] - 1) * B + loc[None, :, None, None]
    gz = (level.map_z.astype(np.int64)[None, None, None, :] - 1) * B + loc[None, None, :, None]
    shape = (B, B, B, level.n_blocks)
    return np.broadcast_to(gx, shape), np.broadcast_to(gy, shape), np.broadcast_to(gz, shape)


def equilibrium(rho, ux, uy, uz, out: Optional[np.ndarray] = None) -> np.ndarray:
    """f_eq(rho,u) per population in Float32 (src/physics_utils.jl:34-39); returns (8,8,8,nb,27) Fortran-ordered.
    The 27 populations are independent: large levels spread them over a few threads (numpy releases the GIL)."""
    rho, ux, uy, uz = (np.asarray(a, dtype=np.float32) for a in (rho, ux, uy, uz))
    if out is None:
        out = np.empty(rho.shape + (27,), dtype=np.float32, order="F")
    usq = ux * ux + uy * uy + uz * uz

    def one(k):
        cu = np.float32(_CX[k]) * ux + np.float32(_CY[k]) * uy + np.float32(_CZ[k]) * uz
        out[..., k] = rho * _W[k] * (np.float32(1) + np.float32(3) * cu + np.float32(4.5) * cu * cu - np.float32(1.5) * usq)

    if rho.size >= (1 << 22):
        import os
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, min(8, (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(one, range(27)))
    else:
        for k in range(27):
            one(k)
    return out


def set_state(level: BlockLevel, rho, ux, uy, uz, f: Optional[np.ndarray] = None, share_ab_buffers: bool = False) -> None:
    """Put one macroscopic state (and matching distributions) into every A/B buffer of the level.

    share_ab_buffers: f_temp / vel_temp become the SAME host arrays as f / vel instead of copies. Only for a host level that
    is uploaded to a device and not stepped on the host (a 256^3 level saves 2.4 GB of first-touch page faults)."""
    level.rho[...] = rho
    if share_ab_buffers:
        level.vel_temp = level.vel
    for a in ((level.vel,) if share_ab_buffers else (level.vel, level.vel_temp)):
        a[..., 0] = ux; a[..., 1] = uy; a[..., 2] = uz
    if f is None:
        f = equilibrium(level.rho, level.vel[..., 0], level.vel[..., 1], level.vel[..., 2], out=level.f)
    else:
        level.f[...] = f
    if share_ab_buffers:
        level.f_temp = level.f
    else:
        level.f_temp[...] = level.f
    if level.f_old.size > 27:
        level.f_old[...] = level.f
        level.rho_old[...] = level.rho
        level.vel_old[...] = level.vel


def init_taylor_green(level: BlockLevel, n_cells: Tuple[int, int, int], u0: float = 0.03, share_ab_buffers: bool = False) -> None:
    """SURVEY section 8(d) initial field: u = u0 (sin X cos Y cos Z, -cos X sin Y cos Z, 0), rho = 1, f = f_eq."""
    gx, gy, gz = global_cell_coords(level)
    X = 2.0 * np.pi * (gx - 0.5) / n_cells[0]
    Y = 2.0 * np.pi * (gy - 0.5) / n_cells[1]
    Z = 2.0 * np.pi * (gz - 0.5) / n_cells[2]
    ux = (u0 * np.sin(X) * np.cos(Y) * np.cos(Z)).astype(np.float32)
    uy = (-u0 * np.cos(X) * np.sin(Y) * np.cos(Z)).astype(np.float32)
    set_state(level, np.float32(1.0), ux, uy, np.float32(0.0), share_ab_buffers=share_ab_buffers)


def init_perturbed(level: BlockLevel, seed: int, u_mean: float = 0.04, amp: float = 0.01) -> None:
    """A deterministic, non-equilibrium, non-uniform state so that every term of the collision is exercised."""
    rng = np.random.default_rng(seed)
    shp = level.rho.shape
    gx, gy, gz = global_cell_coords(level)
    s = 2 ** (level.level_id - 1)
    ux = (u_mean + amp * np.sin(0.37 * gx / s) * np.cos(0.23 * gy / s) + 0.2 * amp * rng.standard_normal(shp)).astype(np.float32)
    uy = (amp * np.cos(0.31 * gx / s) * np.sin(0.29 * gz / s) + 0.2 * amp * rng.standard_normal(shp)).astype(np.float32)
    uz = (amp * np.sin(0.19 * gy / s) * np.sin(0.41 * gz / s) + 0.2 * amp * rng.standard_normal(shp)).astype(np.float32)
    rho = (1.0 + 0.5 * amp * np.cos(0.27 * gx / s + 0.13 * gz / s) + 0.05 * amp * rng.standard_normal(shp)).astype(np.float32)
    f = equilibrium(rho, ux, uy, uz)
    f *= (1.0 + 0.02 * rng.standard_normal(f.shape)).astype(np.float32)
    set_state(level, rho, ux, uy, uz, f)
    # the previous-step velocity buffers hold something slightly different from the moments of f
    level.vel_temp[...] = level.vel * np.float32(0.97)


# ----------------------------------------------------------------------------------------------------------------
# geometry: analytic sphere (stands in for voxelize_blocks!/flood fill/compute_bouzidi_qmap_sparse of row N1)
# ----------------------------------------------------------------------------------------------------------------
def add_sphere(level: BlockLevel, center, radius: float, *, bouzidi: bool = True, wall_dist: bool = True, period=None,
               list_blocks: Optional[int] = None) -> None:
    """Mark cells whose centre lies inside the sphere as obstacle; optionally fill the wall-distance field and build a
    Bouzidi q-map (Float16) + sparse boundary-cell list with the list semantics of src/bouzidi_setup.jl:115-143
    (every cell, solid ones included, one of whose 26 links crosses the surface within the link length).
    period (cells per axis): a lattice of spheres, one per period (multi-GPU loop-back tests: every brick holds the same body);
    list_blocks: only cells of the first so many blocks enter the Bouzidi list (the owned blocks of a rank's view)."""
    gx, gy, gz = global_cell_coords(level)
    px, py, pz = gx - 0.5 - center[0], gy - 0.5 - center[1], gz - 0.5 - center[2]
    if period is not None:
        px, py, pz = [(q + 0.5 * L) % L - 0.5 * L for q, L in zip((px, py, pz), period)]
    r = np.sqrt(px * px + py * py + pz * pz)
    level.obstacle[...] = r < radius
    if wall_dist:
        d = (r - radius).astype(np.float32)
        near = (~level.obstacle) & (d < 4.0)
        level.wall_dist[...] = np.float32(100.0)
        level.wall_dist[near] = np.maximum(d[near], np.float32(0.05))
    if not bouzidi:
        return
    B, n = BLOCK_SIZE, level.n_blocks
    q_map = np.zeros((B, B, B, n, 27), dtype=np.float16, order="F")
    band = np.abs(r - radius) < 2.0
    any_hit = np.zeros((B, B, B, n), dtype=bool)
    for k in range(27):
        c = np.array([_CX[k], _CY[k], _CZ[k]], dtype=np.float64)
        if not c.any():
            continue
        # |p + t c|^2 = R^2, smallest root in (0, 1]
        a = float(c @ c)
        bq = 2.0 * (px * c[0] + py * c[1] + pz * c[2])
        cq = r * r - radius * radius
        disc = bq * bq - 4.0 * a * cq
        ok = band & (disc >= 0.0)
        sq = np.sqrt(np.where(ok, disc, 0.0))
        t1, t2 = (-bq - sq) / (2 * a), (-bq + sq) / (2 * a)
        t = np.where((t1 > 1e-9), t1, t2)
        hit = ok & (t > 1e-9) & (t <= 1.0)
        q_map[..., k][hit] = t[hit].astype(np.float16)     # single rounding Float64 -> Float16 (src/bouzidi_setup.jl:128)
        any_hit |= hit
    if list_blocks is not None:
        any_hit[:, :, :, list_blocks:] = False
    idx = np.argwhere(any_hit)          # rows (x,y,z,b), 0-based
    order = np.lexsort((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]))
    idx = idx[order]
    level.bouzidi_q_map = q_map
    level.bouzidi_cell_x = (idx[:, 0] + 1).astype(np.int8)
    level.bouzidi_cell_y = (idx[:, 1] + 1).astype(np.int8)
    level.bouzidi_cell_z = (idx[:, 2] + 1).astype(np.int8)
    level.bouzidi_cell_block = (idx[:, 3] + 1).astype(np.int32)
    level.n_boundary_cells = int(idx.shape[0])
    level.bouzidi_enabled = level.n_boundary_cells > 0
    if level.bouzidi_enabled and level.f_post_collision.size <= 27:
        level.f_post_collision = np.zeros((B, B, B, n, 27), dtype=np.float32, order="F")


def add_sponge(level: BlockLevel, nx_global: int, thickness_frac: float = 0.25, inlet_frac: float = 0.05) -> None:
    """Cosine outlet/inlet sponge in the spirit of apply_sponge! (src/domain_generation.jl:215-289)."""
    gx, _, _ = global_cell_coords(level)
    x = (gx - 0.5) / nx_global
    sp = np.zeros(level.rho.shape, dtype=np.float64)
    out = x > 1.0 - thickness_frac
    sp[out] = 0.5 * (1 - np.cos(np.pi * (x[out] - (1.0 - thickness_frac)) / thickness_frac))
    inl = x < inlet_frac
    sp[inl] = np.maximum(sp[inl], 0.05 * 0.5 * (1 + np.cos(np.pi * x[inl] / inlet_frac)))
    level.sponge[...] = sp.astype(np.float32)


def refine_region(parent: BlockLevel, lo: Tuple[int, int, int], hi: Tuple[int, int, int], tau: float,
                  temporal: bool = True) -> BlockLevel:
    """Child level covering parent blocks lo..hi (inclusive, 1-based) with all 8 children each
    (child block = 2*parent - 1 + {0,1}, src/domain.jl:100-108)."""
    coords = []
    for cbx in range(lo[0], hi[0] + 1):
        for cby in range(lo[1], hi[1] + 1):
            for cbz in range(lo[2], hi[2] + 1):
                for d in range(8):
                    coords.append((2 * cbx - 1 + (d & 1), 2 * cby - 1 + ((d >> 1) & 1), 2 * cbz - 1 + ((d >> 2) & 1)))
    pdims = (parent.grid_dim_x, parent.grid_dim_y, parent.grid_dim_z)
    dims = tuple(2 * d for d in pdims)
    return make_level(parent.level_id + 1, coords, dims, tau, temporal=temporal)


# ----------------------------------------------------------------------------------------------------------------
# ready-made cases
# ----------------------------------------------------------------------------------------------------------------
def periodic_box(n_blocks_xyz: Tuple[int, int, int], tau: float = 0.5006, u0: float = 0.03, init: bool = True, upload_only: bool = False):
    """SURVEY section 8(d) C1/C2 workload: uniform periodic box (wrapped neighbor_table), Taylor-Green start.
    upload_only: the host level will only be uploaded to a device (see set_state's share_ab_buffers)."""
    nbx, nby, nbz = n_blocks_xyz
    level = make_level(1, full_box_coords(nbx, nby, nbz), (nbx, nby, nbz), tau, periodic=(True, True, True), temporal=False)
    if init:
        init_taylor_green(level, (nbx * 8, nby * 8, nbz * 8), u0, share_ab_buffers=upload_only)
    params = SolverParams(domain_nx=nbx * 8, domain_ny=nby * 8, domain_nz=nbz * 8, wall_model_active=False, c_wale=0.5,
                          nu_sgs_bg=0.0005, inlet_turbulence=0.0, use_temporal_interp=False, sponge_blend_dist=False)
    return [level], params


def tunnel_with_sphere(n_blocks_xyz=(6, 4, 4), *, tau: float = 0.5006, levels: int = 1, wall_model: bool = False,
                       bouzidi: bool = True, sponge_blend: bool = True, symmetric: bool = False,
                       inlet_turbulence: float = 0.01, seed: int = 7, temporal: bool = True):
    """A small wind tunnel (inlet/outlet/mirror edges) with a sphere, sponge, optional wall model, Bouzidi on the
    finest level and optional nested refinement around the body - every branch of the hot path in one case."""
    nbx, nby, nbz = n_blocks_xyz
    grids: List[BlockLevel] = []
    l1 = make_level(1, full_box_coords(nbx, nby, nbz), (nbx, nby, nbz), tau, temporal=temporal)
    grids.append(l1)
    center1 = np.array([nbx * 8 * 0.4, nby * 8 * (0.0 if symmetric else 0.5), nbz * 8 * 0.5])
    radius1 = 0.9 * 8 * min(nby, nbz) / 6.0 + 2.0
    cb = np.floor(center1 / 8).astype(int) + 1
    lo = tuple(int(max(1, cb[i] - 1)) for i in range(3))
    hi = tuple(int(min((nbx, nby, nbz)[i], cb[i] + 1)) for i in range(3))
    for lvl in range(2, levels + 1):
        tau_l = 0.5 + (tau - 0.5) / 2 ** (lvl - 1)     # as written in src/physics_scaling.jl:139-143: tau-1/2 halves per level
        child = refine_region(grids[-1], lo, hi, tau_l, temporal=temporal)
        grids.append(child)
        # next refinement: the central part of this one
        clo = tuple(2 * l - 1 for l in lo); chi = tuple(2 * h for h in hi)
        lo = tuple(clo[i] + 1 for i in range(3)); hi = tuple(chi[i] - 1 for i in range(3))
    for i, g in enumerate(grids):
        s = 2 ** i
        add_sphere(g, center1 * s, radius1 * s, bouzidi=(bouzidi and i == len(grids) - 1), wall_dist=wall_model)
        add_sponge(g, nbx * 8 * s)
        init_perturbed(g, seed + i)
    params = SolverParams(domain_nx=nbx * 8, domain_ny=nby * 8, domain_nz=nbz * 8, wall_model_active=wall_model,
                          c_wale=0.5, nu_sgs_bg=0.0005, inlet_turbulence=inlet_turbulence, use_temporal_interp=temporal,
                          sponge_blend_dist=sponge_blend, symmetric_analysis=symmetric, q_min_threshold=0.001)
    return grids, params
