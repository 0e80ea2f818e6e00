"""Host pre-processing (scope row N1): YAML + STL -> BlockLevels, restating the reference's one-time setup.

  config keys        src/config_loader.jl:109-197 (Appendix C of SURVEY.md)
  scaling            src/physics_scaling.jl:59-176
  STL                src/geometry.jl:86-209
  block sets         src/domain.jl:56-247, src/domain_topology.jl:9-133
  voxelizer / fill   src/domain_generation.jl:10-203
  sponge             src/domain_generation.jl:205-289
  wall distance      src/domain_generation.jl:371-434
  Bouzidi q-map      src/bouzidi_setup.jl:64-167, src/bouzidi_math.jl:9-102

All of it is Float64 host arithmetic that runs once per case; it is numpy here. Where the reference loops over cells and
triangles, the loops are vectorised with conservative pre-filters that cannot change a result (noted in place).
Anchors: the setup integers printed in the reference's run logs (tests/golden/sphere_re266k_setup.json).
"""
from __future__ import annotations

import math
import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np
import yaml

from .blocks import BLOCK_SIZE, BlockLevel, build_neighbor_table
from .physics import SolverParams

BS = BLOCK_SIZE


# ----------------------------------------------------------------------------------------------------------------
# configuration, src/config_loader.jl:109-197
# ----------------------------------------------------------------------------------------------------------------
def _get(d, *keys, default=None, required=False):
    cur = d
    for k in keys:
        if not isinstance(cur, dict) or k not in cur:
            if required:
                raise KeyError("Missing config key: " + " -> ".join(keys))
            return default
        cur = cur[k]
    return default if (cur is None and default is not None) else cur


@dataclass
class CaseConfig:
    stl_file: str
    stl_scale: float
    surface_resolution: int
    num_levels: int
    output_dir: str
    steps: int
    ramp_steps: int
    output_freq: int
    symmetric_analysis: bool = False
    reference_area_full_model: float = 0.0
    reference_chord: float = 0.0
    reference_length_for_meshing: float = 0.0
    reference_dimension: str = "x"
    fluid_density: float = 1.225
    fluid_kinematic_viscosity: float = 1.5e-5
    flow_velocity: float = 10.0
    u_lattice: np.float32 = np.float32(0.01)
    c_wale: np.float32 = np.float32(0.20)
    tau_min: np.float32 = np.float32(0.505)
    inlet_turbulence_intensity: np.float32 = np.float32(0.01)
    nu_sgs_background: np.float32 = np.float32(0.0005)
    sponge_blend_distributions: bool = True
    temporal_interpolation: bool = True
    auto_levels: bool = False
    max_levels: int = 12
    min_coarse_blocks: int = 4
    wall_model_enabled: bool = False
    domain_upstream: float = 0.75
    domain_downstream: float = 1.5
    domain_lateral: float = 0.75
    domain_height: float = 0.75
    sponge_thickness: np.float32 = np.float32(0.10)
    block_size_config: int = 8
    refinement_margin: int = 2
    refinement_strategy: str = "geometry_first"
    wake_enabled: bool = False
    wake_length: float = 0.25
    wake_width_factor: float = 0.1
    wake_height_factor: float = 0.1
    boundary_method: str = "bouzidi"
    bouzidi_levels: int = 1
    q_min_threshold: np.float32 = np.float32(0.001)
    forces_enabled: bool = True
    moment_center: Tuple[float, float, float] = (0.25, 0.0, 0.0)
    diag_freq: int = 500
    async_depth: int = 8
    case_dir: str = ""
    output_fields: Tuple[str, ...] = ("Density", "Velocity", "VelocityMagnitude", "Obstacle", "Level")   # io_vtk.jl:116-120

    @property
    def reference_area_config(self) -> float:
        return self.reference_area_full_model / 2.0 if self.symmetric_analysis else self.reference_area_full_model


def load_case_configuration(config_path: str, overrides: Optional[dict] = None) -> CaseConfig:
    """load_case_configuration (src/config_loader.jl:109-197). `overrides` patches the parsed YAML tree, e.g.
    {"basic": {"surface_resolution": 25, "flow": {"velocity": 4.0}}} (how the Re 266k log was produced)."""
    with open(config_path) as fh:
        cfg = yaml.safe_load(fh)

    def merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                merge(dst[k], v)
            else:
                dst[k] = v
    if overrides:
        merge(cfg, overrides)
    f32 = np.float32
    g = lambda *k, **kw: _get(cfg, *k, **kw)
    return CaseConfig(
        stl_file=g("basic", "stl_file", required=True), stl_scale=float(g("basic", "stl_scale", required=True)),
        surface_resolution=int(g("basic", "surface_resolution", required=True)), num_levels=int(g("basic", "num_levels", required=True)),
        output_dir=g("basic", "simulation", "output_dir", required=True), steps=int(g("basic", "simulation", "steps", required=True)),
        ramp_steps=int(g("basic", "simulation", "ramp_steps", required=True)), output_freq=int(g("basic", "simulation", "output_freq", required=True)),
        symmetric_analysis=bool(g("advanced", "refinement", "symmetric_analysis", default=False)),
        reference_area_full_model=float(g("basic", "reference_area_of_full_model", default=0.0)),
        reference_chord=float(g("basic", "reference_chord", default=0.0)),
        reference_length_for_meshing=float(g("basic", "reference_length_for_meshing", default=0.0)),
        reference_dimension=str(g("basic", "reference_dimension", default="x")),
        fluid_density=float(g("basic", "fluid", "density", default=1.225)),
        fluid_kinematic_viscosity=float(g("basic", "fluid", "kinematic_viscosity", default=1.5e-5)),
        flow_velocity=float(g("basic", "flow", "velocity", default=10.0)),
        u_lattice=f32(g("advanced", "numerics", "u_lattice", default=0.01)), c_wale=f32(g("advanced", "numerics", "c_wale", default=0.20)),
        tau_min=f32(g("advanced", "numerics", "tau_min", default=0.505)),
        inlet_turbulence_intensity=f32(g("advanced", "numerics", "inlet_turbulence_intensity", default=0.01)),
        nu_sgs_background=f32(g("advanced", "numerics", "nu_sgs_background", default=0.0005)),
        sponge_blend_distributions=bool(g("advanced", "numerics", "sponge_blend_distributions", default=True)),
        temporal_interpolation=bool(g("advanced", "numerics", "temporal_interpolation", default=True)),
        auto_levels=bool(g("advanced", "high_re", "auto_levels", default=False)), max_levels=int(g("advanced", "high_re", "max_levels", default=12)),
        min_coarse_blocks=int(g("advanced", "high_re", "min_coarse_blocks", default=4)),
        wall_model_enabled=bool(g("advanced", "high_re", "wall_model", "enabled", default=False)),
        domain_upstream=float(g("advanced", "domain", "upstream", default=0.75)), domain_downstream=float(g("advanced", "domain", "downstream", default=1.5)),
        domain_lateral=float(g("advanced", "domain", "lateral", default=0.75)), domain_height=float(g("advanced", "domain", "height", default=0.75)),
        sponge_thickness=f32(g("advanced", "domain", "sponge_thickness", default=0.10)),
        block_size_config=int(g("advanced", "refinement", "block_size", default=8)), refinement_margin=int(g("advanced", "refinement", "margin", default=2)),
        refinement_strategy=str(g("advanced", "refinement", "strategy", default="geometry_first")),
        wake_enabled=bool(g("advanced", "refinement", "wake_enabled", default=False)), wake_length=float(g("advanced", "refinement", "wake_length", default=0.25)),
        wake_width_factor=float(g("advanced", "refinement", "wake_width_factor", default=0.1)),
        wake_height_factor=float(g("advanced", "refinement", "wake_height_factor", default=0.1)),
        boundary_method=str(g("advanced", "boundary", "method", default="bouzidi")), bouzidi_levels=int(g("advanced", "boundary", "bouzidi_levels", default=1)),
        q_min_threshold=f32(g("advanced", "boundary", "q_min_threshold", default=0.001)),
        forces_enabled=bool(g("advanced", "forces", "enabled", default=True)),
        moment_center=tuple(float(v) for v in g("advanced", "forces", "moment_center", default=[0.25, 0.0, 0.0])),
        diag_freq=int(g("advanced", "diagnostics", "freq", default=500)), async_depth=int(g("advanced", "gpu", "async_depth", default=8)),
        case_dir=os.path.dirname(os.path.abspath(config_path)),
        # vorticity / bouzidi are read by the reference's loader (config_loader.jl:145,148) but never written by io_vtk.jl
        output_fields=tuple(name for key, name in (("density", "Density"), ("velocity", "Velocity"), ("velocity_magnitude", "VelocityMagnitude"),
                                                   ("obstacle", "Obstacle"), ("level", "Level"))
                            if bool(g("basic", "simulation", "output_fields", key, default=True))),
    )


# ----------------------------------------------------------------------------------------------------------------
# geometry, src/geometry.jl
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class SolverMesh:
    triangles: np.ndarray      # [n,3,3] Float64 (Float32 file values widened, times scale)
    min_bounds: np.ndarray
    max_bounds: np.ndarray
    normals: np.ndarray        # [n,3]
    areas: np.ndarray          # [n]
    centers: np.ndarray        # [n,3]


def load_mesh(filename: str, scale: float = 1.0) -> SolverMesh:
    """load_mesh (src/geometry.jl:160-209): binary/ASCII sniffing, Float32 -> Float64 * scale, properties from vertices."""
    size = os.path.getsize(filename)
    is_binary = True
    with open(filename, "rb") as fh:
        if size < 84:
            is_binary = False
        else:
            head = fh.read(5)
            if head.decode("latin1").lower().startswith("solid"):
                fh.seek(80)
                count = struct.unpack("<I", fh.read(4))[0]
                if size != 84 + count * 50:
                    is_binary = False
    if is_binary:
        raw = open(filename, "rb").read()
        count = struct.unpack("<I", raw[80:84])[0]
        rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=count, offset=84)
        tris = rec["v"].astype(np.float64) * scale
    else:
        verts, cur = [], []
        for line in open(filename):
            s = line.strip()
            if s.startswith("vertex"):
                p = s.split()
                if len(p) >= 4:
                    cur.append([float(p[1]) * scale, float(p[2]) * scale, float(p[3]) * scale])
            elif s.startswith("endloop"):
                if len(cur) == 3:
                    verts.append(cur)
                cur = []
        tris = np.asarray(verts, dtype=np.float64).reshape(-1, 3, 3)
    if tris.shape[0] == 0:
        raise ValueError("No triangles loaded.")
    e1, e2 = tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    cp = np.cross(e1, e2)
    area = 0.5 * np.linalg.norm(cp, axis=1)
    normal = np.where((area > 1e-12)[:, None], cp / np.maximum(2.0 * area, 1e-300)[:, None], 0.0)
    centers = (tris[:, 0] + tris[:, 1] + tris[:, 2]) / 3.0
    flat = tris.reshape(-1, 3)
    return SolverMesh(tris, flat.min(axis=0), flat.max(axis=0), normal, area, centers)


# ----------------------------------------------------------------------------------------------------------------
# scaling, src/physics_scaling.jl:59-176
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class DomainParameters:
    num_levels: int
    mesh_min: np.ndarray
    mesh_max: np.ndarray
    mesh_center: np.ndarray
    mesh_extent: np.ndarray
    reference_length: float
    reference_chord: float
    reference_area: float
    moment_center: Tuple[float, float, float]
    domain_size: Tuple[float, float, float]
    mesh_offset: np.ndarray
    dx_fine: float
    dx_coarse: float
    dx_levels: List[float]
    nx_coarse: int
    ny_coarse: int
    nz_coarse: int
    bx_max: int
    by_max: int
    bz_max: int
    tau_levels: List[np.float32]
    re_number: float
    u_physical: float
    rho_physical: float
    nu_physical: float
    length_scale: float
    time_scale: float
    velocity_scale: float
    force_scale: float
    tau_fine: float
    wall_model_active: bool


def compute_domain_from_mesh(mesh_min, mesh_max, cfg: CaseConfig) -> DomainParameters:
    mesh_min = np.asarray(mesh_min, dtype=np.float64)
    mesh_max = np.asarray(mesh_max, dtype=np.float64)
    center = (mesh_min + mesh_max) / 2
    extent = mesh_max - mesh_min
    if cfg.reference_length_for_meshing > 0:
        L = cfg.reference_length_for_meshing
    else:
        L = {"x": extent[0], "y": extent[1], "z": extent[2]}.get(cfg.reference_dimension, float(extent.max()))
    chord = cfg.reference_chord if cfg.reference_chord > 0 else float(extent[0])
    area = cfg.reference_area_config if cfg.reference_area_config > 0 else (
        extent[1] * extent[2] * 2 if cfg.symmetric_analysis else extent[1] * extent[2])
    u_phys, nu_phys, rho_phys = cfg.flow_velocity, cfg.fluid_kinematic_viscosity, cfg.fluid_density
    re = u_phys * L / nu_phys
    tau_fine_computed = 3.0 * (float(cfg.u_lattice) * cfg.surface_resolution / re) + 0.5        # :66-69
    tau_fine = max(tau_fine_computed, float(cfg.tau_min))
    dom_x = L * (cfg.domain_upstream + cfg.domain_downstream) + extent[0]
    dom_y = (mesh_max[1] + L * cfg.domain_lateral) if cfg.symmetric_analysis else (extent[1] + 2 * L * cfg.domain_lateral)
    dom_z = extent[2] + 2 * L * cfg.domain_height
    dx_fine = L / cfg.surface_resolution
    min_domain = min(dom_x, dom_y, dom_z)
    ratio = min_domain / (dx_fine * cfg.min_coarse_blocks * cfg.block_size_config)
    max_levels_domain = 1 if ratio < 1.0 else int(math.floor(1 + math.log2(ratio)))
    if cfg.num_levels > 0:
        num_levels = min(cfg.num_levels, max_levels_domain)
    else:
        num_levels = min(max_levels_domain, cfg.max_levels) if cfg.auto_levels else min(8, max_levels_domain)
    dx_coarse = dx_fine * 2 ** (num_levels - 1)
    dx_levels = [dx_fine * 2 ** (num_levels - lvl) for lvl in range(1, num_levels + 1)]
    B = cfg.block_size_config
    ncoarse = lambda d: max(B, int(math.ceil(math.ceil(d / dx_coarse) / B) * B))
    nx, ny, nz = ncoarse(dom_x), ncoarse(dom_y), ncoarse(dom_z)
    dom_x, dom_y, dom_z = nx * dx_coarse, ny * dx_coarse, nz * dx_coarse
    mesh_x = L * cfg.domain_upstream
    mesh_y = 0.0 if cfg.symmetric_analysis else (dom_y / 2 - center[1])
    mesh_z = dom_z / 2 - center[2]
    offset = np.array([mesh_x - mesh_min[0], mesh_y, mesh_z])
    length_scale = dx_fine
    velocity_scale = u_phys / float(cfg.u_lattice)
    time_scale = length_scale / velocity_scale
    tau_levels = [np.float32(tau_fine if lvl == num_levels else 0.5 + (tau_fine - 0.5) * 2.0 ** (num_levels - lvl))
                  for lvl in range(1, num_levels + 1)]
    force_scale = rho_phys * length_scale ** 4 / time_scale ** 2
    mc = cfg.moment_center
    moment_center = (mesh_min[0] + offset[0] + mc[0] * chord, center[1] + offset[1] + mc[1] * chord, center[2] + offset[2] + mc[2] * chord)
    return DomainParameters(num_levels, mesh_min, mesh_max, center, extent, L, chord, float(area), moment_center, (dom_x, dom_y, dom_z),
                            offset, dx_fine, dx_coarse, dx_levels, nx, ny, nz, nx // B, ny // B, nz // B, tau_levels, re, u_phys,
                            rho_phys, nu_phys, length_scale, time_scale, velocity_scale, force_scale, tau_fine, cfg.wall_model_enabled)


# ----------------------------------------------------------------------------------------------------------------
# block topology, src/domain_topology.jl:9-133
# ----------------------------------------------------------------------------------------------------------------
Coord = Tuple[int, int, int]


def get_active_blocks_for_level(mesh: SolverMesh, dx: float, offset, bmax: Coord) -> Set[Coord]:
    margin = dx * 0.01
    inv = 1.0 / (BS * dx)
    t = mesh.triangles + np.asarray(offset)
    tmin, tmax = t.min(axis=1), t.max(axis=1)
    lo = np.floor((tmin - margin) * inv).astype(np.int64) + 1
    hi = np.floor((tmax + margin) * inv).astype(np.int64) + 1
    lo = np.maximum(lo, 1)
    hi = np.minimum(hi, np.asarray(bmax))
    out: Set[Coord] = set()
    for (a, b) in {(tuple(l), tuple(h)) for l, h in zip(lo, hi)}:
        for bz in range(a[2], b[2] + 1):
            for by in range(a[1], b[1] + 1):
                for bx in range(a[0], b[0] + 1):
                    out.add((bx, by, bz))
    return out


def _siblings(c: Coord):
    p = ((c[0] + 1) // 2, (c[1] + 1) // 2, (c[2] + 1) // 2)
    for dbz in (0, 1):
        for dby in (0, 1):
            for dbx in (0, 1):
                yield (2 * p[0] - 1 + dbx, 2 * p[1] - 1 + dby, 2 * p[2] - 1 + dbz)


def _inside(c: Coord, bmax: Coord) -> bool:
    return 1 <= c[0] <= bmax[0] and 1 <= c[1] <= bmax[1] and 1 <= c[2] <= bmax[2]


def add_halo_blocks_with_siblings(active: Set[Coord], layers: int, bmax: Coord) -> None:
    offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy, dz) != (0, 0, 0)]
    for _ in range(layers):
        new = set()
        for (bx, by, bz) in active:
            for (dx, dy, dz) in offs:
                n = (bx + dx, by + dy, bz + dz)
                if _inside(n, bmax) and n not in active:
                    new.add(n)
        sib = set()
        for c in new:
            for s in _siblings(c):
                if _inside(s, bmax) and s not in active and s not in new:
                    sib.add(s)
        active |= new
        active |= sib


def ensure_complete_parent_coverage(active: Set[Coord], bmax: Coord) -> None:
    added, it = True, 0
    while added and it < 10:
        added = False
        it += 1
        sib = set()
        for c in active:
            for s in _siblings(c):
                if _inside(s, bmax) and s not in active:
                    sib.add(s)
                    added = True
        active |= sib


# ----------------------------------------------------------------------------------------------------------------
# dense helper: a level's cells as one boolean volume (blocks that do not exist are "inactive")
# ----------------------------------------------------------------------------------------------------------------
class _Dense:
    def __init__(self, coords: Sequence[Coord]):
        c = np.asarray(coords, dtype=np.int64)
        self.lo = c.min(axis=0)
        self.nb = c.max(axis=0) - self.lo + 1
        self.rel = c - self.lo
        self.shape = tuple(int(v) for v in self.nb * BS)
        self.active = np.zeros(self.shape, dtype=bool)
        for (x, y, z) in self.rel:
            self.active[x * BS:(x + 1) * BS, y * BS:(y + 1) * BS, z * BS:(z + 1) * BS] = True

    def to_dense(self, arr4: np.ndarray) -> np.ndarray:
        out = np.zeros(self.shape, dtype=arr4.dtype)
        for i, (x, y, z) in enumerate(self.rel):
            out[x * BS:(x + 1) * BS, y * BS:(y + 1) * BS, z * BS:(z + 1) * BS] = arr4[:, :, :, i]
        return out

    def to_blocks(self, dense: np.ndarray, n: int) -> np.ndarray:
        out = np.zeros((BS, BS, BS, n), dtype=dense.dtype, order="F")
        for i, (x, y, z) in enumerate(self.rel):
            out[:, :, :, i] = dense[x * BS:(x + 1) * BS, y * BS:(y + 1) * BS, z * BS:(z + 1) * BS]
        return out


def _cell_centers(coords, dx: float):
    """cell centres ((b-1)*8 + l - 0.5) dx, arrays of shape (8,8,8,nb) (src/domain_generation.jl:91-93)"""
    c = np.asarray(coords, dtype=np.float64)
    l = np.arange(1, BS + 1, dtype=np.float64)
    px = ((c[:, 0][None, None, None, :] - 1) * BS + l[:, None, None, None] - 0.5) * dx
    py = ((c[:, 1][None, None, None, :] - 1) * BS + l[None, :, None, None] - 0.5) * dx
    pz = ((c[:, 2][None, None, None, :] - 1) * BS + l[None, None, :, None] - 0.5) * dx
    shp = (BS, BS, BS, len(coords))
    return np.broadcast_to(px, shp), np.broadcast_to(py, shp), np.broadcast_to(pz, shp)


def _block_triangle_map(mesh: SolverMesh, coords, dx: float, offset, margin: float) -> List[np.ndarray]:
    """build_block_triangle_map (src/domain_generation.jl:34-72) / ..._for_bouzidi (src/bouzidi_setup.jl:11-52)"""
    lookup = {tuple(c): i for i, c in enumerate(coords)}
    t = mesh.triangles + np.asarray(offset)
    lo = np.floor((t.min(axis=1) - margin) / (BS * dx)).astype(np.int64) + 1
    hi = np.floor((t.max(axis=1) + margin) / (BS * dx)).astype(np.int64) + 1
    lo = np.maximum(lo, 1)
    lists: List[List[int]] = [[] for _ in coords]
    for ti in range(t.shape[0]):
        for bz in range(lo[ti, 2], hi[ti, 2] + 1):
            for by in range(lo[ti, 1], hi[ti, 1] + 1):
                for bx in range(lo[ti, 0], hi[ti, 0] + 1):
                    i = lookup.get((bx, by, bz))
                    if i is not None:
                        lists[i].append(ti)
    return [np.asarray(l, dtype=np.int64) for l in lists]


# ----------------------------------------------------------------------------------------------------------------
# voxelizer, src/domain_generation.jl:10-112
# ----------------------------------------------------------------------------------------------------------------
def _tri_box_overlap(centers: np.ndarray, half: float, v1: np.ndarray, v2: np.ndarray, v3: np.ndarray) -> np.ndarray:
    """triangle_intersects_aabb for every (cell, triangle) pair: AABB slabs + the 9 edge-cross axes (the reference has
    no triangle-plane test). centers [C,3]; v* [T,3]; returns bool [C,T]."""
    h = half * 1.001
    t1 = v1[None, :, :] - centers[:, None, :]
    t2 = v2[None, :, :] - centers[:, None, :]
    t3 = v3[None, :, :] - centers[:, None, :]
    mn = np.minimum(np.minimum(t1, t2), t3)
    mx = np.maximum(np.maximum(t1, t2), t3)
    ok = ~((mn > h).any(axis=2) | (mx < -h).any(axis=2))
    if not ok.any():
        return ok
    f = [t2 - t1, t3 - t2, t1 - t3]
    for i in range(3):
        u = np.zeros(3)
        u[i] = 1.0
        for j in range(3):
            axis = np.cross(u, f[j])
            use = (axis * axis).sum(axis=2) >= 1e-10
            p1 = (t1 * axis).sum(axis=2); p2 = (t2 * axis).sum(axis=2); p3 = (t3 * axis).sum(axis=2)
            aa = np.abs(axis)
            r = (h * aa[..., 0] + h * aa[..., 1]) + h * aa[..., 2]          # :28, term by term
            sep = (np.minimum(p1, np.minimum(p2, p3)) > r) | (np.maximum(p1, np.maximum(p2, p3)) < -r)
            ok &= ~(use & sep)
    return ok


def _candidate_pairs(tmin: np.ndarray, tmax: np.ndarray, reach: float, dx: float, dense: "_Dense"):
    """All (cell, triangle) pairs whose cell CENTRE lies within `reach` of the triangle's bounding box (per axis), for
    cells of existing blocks. Returns (flat dense cell index, triangle index, cell centre [P,3])."""
    org = dense.lo.astype(np.float64) - 1.0                     # block coord 1 starts at cell 0
    lo = np.ceil((tmin - reach) / dx - 0.5 - 1e-9).astype(np.int64) - (org * BS).astype(np.int64)
    hi = np.floor((tmax + reach) / dx - 0.5 + 1e-9).astype(np.int64) - (org * BS).astype(np.int64)
    shp = np.asarray(dense.shape)
    lo = np.maximum(lo, 0)
    hi = np.minimum(hi, shp - 1)
    ext = np.maximum(hi - lo + 1, 0)
    cnt = ext[:, 0] * ext[:, 1] * ext[:, 2]
    tri = np.repeat(np.arange(len(cnt)), cnt)
    start = np.repeat(np.cumsum(cnt) - cnt, cnt)
    r = np.arange(cnt.sum()) - start
    ex, ey = ext[tri, 0], ext[tri, 1]
    ix = lo[tri, 0] + r % ex
    iy = lo[tri, 1] + (r // ex) % ey
    iz = lo[tri, 2] + r // (ex * ey)
    keep = dense.active[ix, iy, iz]
    ix, iy, iz, tri = ix[keep], iy[keep], iz[keep], tri[keep]
    base = (org * BS)
    centre = np.stack([(ix + base[0] + 0.5) * dx, (iy + base[1] + 0.5) * dx, (iz + base[2] + 0.5) * dx], axis=1)
    return np.ravel_multi_index((ix, iy, iz), dense.shape), tri, centre


def _tri_box_overlap_pairs(c: np.ndarray, half: float, v1: np.ndarray, v2: np.ndarray, v3: np.ndarray) -> np.ndarray:
    """triangle_intersects_aabb (src/domain_generation.jl:10-32) for a list of (cell centre, triangle) pairs [P,3]:
    AABB slabs + the 9 edge-cross axes (the reference has no triangle-plane test)."""
    h = half * 1.001
    t1, t2, t3 = v1 - c, v2 - c, v3 - c
    mn = np.minimum(np.minimum(t1, t2), t3)
    mx = np.maximum(np.maximum(t1, t2), t3)
    ok = ~((mn > h).any(axis=1) | (mx < -h).any(axis=1))
    f = [t2 - t1, t3 - t2, t1 - t3]
    for i in range(3):
        u = np.zeros(3)
        u[i] = 1.0
        for j in range(3):
            axis = np.cross(u, f[j])
            use = (axis * axis).sum(axis=1) >= 1e-10
            p1 = (t1 * axis).sum(axis=1); p2 = (t2 * axis).sum(axis=1); p3 = (t3 * axis).sum(axis=1)
            aa = np.abs(axis)
            r = (h * aa[:, 0] + h * aa[:, 1]) + h * aa[:, 2]                # :28, term by term
            sep = (np.minimum(p1, np.minimum(p2, p3)) > r) | (np.maximum(p1, np.maximum(p2, p3)) < -r)
            ok &= ~(use & sep)
    return ok


def _native_inputs(coords, mesh: SolverMesh, offset):
    """(library, int32 [nb,3] block coordinates, Float64 [T,3,3] triangles with the mesh offset added, threads)"""
    from . import _setup_lib
    lib = _setup_lib.load()
    c = np.ascontiguousarray(np.asarray(coords, dtype=np.int32).reshape(-1, 3))
    tri = None if mesh is None else np.ascontiguousarray(mesh.triangles + np.asarray(offset), dtype=np.float64)
    return lib, c, tri, _setup_lib.n_threads()


def _ptr(a: np.ndarray):
    return a.ctypes.data


def voxelize_blocks(coords, mesh: SolverMesh, dx: float, offset, method: str = "native") -> np.ndarray:
    """voxelize_blocks! (src/domain_generation.jl:74-112). The reference tests every cell of a block against the
    triangles binned to that block (margin 2 dx); a triangle can only touch a cell's 0.75 dx half-box if the cell centre
    is within 0.75075 dx of the triangle's bounding box, so testing exactly those pairs gives the same voxels.
    method "native": libludwig_setup.so, block by block like the reference; "numpy": the restatement below (the checker)."""
    n = len(coords)
    if method == "native":
        from . import _setup_lib
        lib, c, tri, nt = _native_inputs(coords, mesh, offset)
        obs = np.zeros((BS, BS, BS, n), dtype=np.uint8, order="F")
        _setup_lib.check(lib.lws_voxelize(_ptr(tri), tri.shape[0], float(dx), _ptr(c), n, _ptr(obs), nt), "lws_voxelize")
        return obs.view(np.bool_)
    d = _Dense(coords)
    tri = mesh.triangles + np.asarray(offset)
    cell, ti, centre = _candidate_pairs(tri.min(axis=1), tri.max(axis=1), 0.75 * dx * 1.001, dx, d)
    hit = _tri_box_overlap_pairs(centre, 0.75 * dx, tri[ti, 0], tri[ti, 1], tri[ti, 2])
    dense = np.zeros(d.shape, dtype=bool)
    dense.reshape(-1)[cell[hit]] = True
    return d.to_blocks(dense, n)


def perform_flood_fill(obstacle: np.ndarray, coords, method: str = "native") -> int:
    """6-connected fill of everything not reachable from the fluid cells of the min-x blocks (src/domain_generation.jl:114-203).
    Returns the number of interior voxels turned solid (the count the reference prints)."""
    if method == "native":
        from . import _setup_lib
        lib, c, _, _ = _native_inputs(coords, None, None)
        assert obstacle.flags.f_contiguous and obstacle.dtype == np.bool_
        return _setup_lib.check(lib.lws_flood_fill(_ptr(c), len(coords), _ptr(obstacle)), "lws_flood_fill")
    from scipy import ndimage
    d = _Dense(coords)
    obs = d.to_dense(obstacle)
    free = d.active & ~obs
    lab, _ = ndimage.label(free, structure=ndimage.generate_binary_structure(3, 1))
    c = np.asarray(coords)
    seed = np.zeros(d.shape, dtype=bool)
    for (x, y, z) in d.rel[c[:, 0] == c[:, 0].min()]:
        seed[x * BS:(x + 1) * BS, y * BS:(y + 1) * BS, z * BS:(z + 1) * BS] = True
    reach_labels = np.unique(lab[seed & free])
    reach_labels = reach_labels[reach_labels > 0]
    visited = np.isin(lab, reach_labels)
    fill = free & ~visited
    obstacle[...] = d.to_blocks(obs | fill, len(coords))
    return int(fill.sum())


# ----------------------------------------------------------------------------------------------------------------
# sponge, src/domain_generation.jl:205-289
# ----------------------------------------------------------------------------------------------------------------
def _profile(x: np.ndarray, thickness: float) -> np.ndarray:
    return np.where(x <= 0.0, 1.0, np.where(x >= thickness, 0.0, 0.5 * (1.0 + np.cos(np.pi * x / thickness))))


def apply_sponge(coords, params: DomainParameters, lvl_scale: int, cfg: CaseConfig) -> np.ndarray:
    dx = params.dx_coarse / lvl_scale
    Lx, Ly, Lz = params.domain_size
    st = float(cfg.sponge_thickness)
    outlet_t = Lx * max(st, 0.15)
    inlet_t = Lx * 0.02
    y_t, z_t = Ly * st * 0.5, Lz * st * 0.5
    outlet_start, y_top, z_back = Lx - outlet_t, Ly - y_t, Lz - z_t
    # every term depends on ONE coordinate and the result is their maximum: evaluate the profiles on the [8, nb] coordinate arrays
    # of each axis and combine by broadcasting (same expressions on the same Float64 inputs as the cell-by-cell form)
    c = np.asarray(coords, dtype=np.float64)
    l = np.arange(1, BS + 1, dtype=np.float64)
    px, py, pz = (((c[:, a][None, :] - 1) * BS + l[:, None] - 0.5) * dx for a in range(3))
    vx = np.zeros(px.shape)
    vx = np.where(px > outlet_start, np.maximum(vx, _profile(outlet_t - (px - outlet_start), outlet_t) * 1.0), vx)
    vx = np.where(px < inlet_t, np.maximum(vx, _profile(px, inlet_t) * 0.05), vx)
    vy = np.zeros(py.shape)
    if not cfg.symmetric_analysis:
        vy = np.where(py < y_t, np.maximum(vy, _profile(py, y_t) * 0.1), vy)
    vy = np.where(py > y_top, np.maximum(vy, _profile(y_t - (py - y_top), y_t) * 0.1), vy)
    vz = np.zeros(pz.shape)
    vz = np.where(pz < z_t, np.maximum(vz, _profile(pz, z_t) * 0.1), vz)
    vz = np.where(pz > z_back, np.maximum(vz, _profile(z_t - (pz - z_back), z_t) * 0.1), vz)
    vx, vy, vz = (v.astype(np.float32) for v in (vx, vy, vz))          # rounding is monotone: max then round = round then max
    # built as C-ordered [nb, z, y, x] (memory order of the Fortran [x, y, z, nb] array the level holds), returned as that view
    val = np.maximum(np.maximum(vx.T[:, None, None, :], vy.T[:, None, :, None]), vz.T[:, :, None, None])
    return val.transpose(3, 2, 1, 0)


# ----------------------------------------------------------------------------------------------------------------
# wall distance, src/domain_generation.jl:371-434 (values are metres used as lattice units downstream: Appendix A.6)
# ----------------------------------------------------------------------------------------------------------------
def compute_wall_distances(coords, obstacle: np.ndarray, dx: float, method: str = "native") -> np.ndarray:
    if method == "native":
        from . import _setup_lib
        lib, c, _, nt = _native_inputs(coords, None, None)
        assert obstacle.flags.f_contiguous and obstacle.dtype == np.bool_
        wall = np.full(obstacle.shape, np.float32(100.0), dtype=np.float32, order="F")
        _setup_lib.check(lib.lws_wall_distance(_ptr(c), len(coords), _ptr(obstacle), float(dx), _ptr(wall), nt), "lws_wall_distance")
        return wall
    d = _Dense(coords)
    obs = d.to_dense(obstacle)
    best = np.full(d.shape, np.float32(100.0), dtype=np.float32)
    near = np.zeros(d.shape, dtype=bool)
    pad_o = np.pad(obs, 1)
    fdx = np.float32(dx)
    sx, sy, sz = d.shape
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dxo in (-1, 0, 1):
                if (dxo, dy, dz) == (0, 0, 0):
                    continue
                nb_obs = pad_o[1 + dxo:1 + dxo + sx, 1 + dy:1 + dy + sy, 1 + dz:1 + dz + sz]   # missing blocks hold False
                dist = np.float32(np.sqrt(np.float32(dxo * dxo + dy * dy + dz * dz))) * fdx
                near |= nb_obs
                best = np.where(nb_obs, np.minimum(best, dist), best)
    res = np.where(near & ~obs & d.active, best, np.float32(100.0)).astype(np.float32)
    return d.to_blocks(res, len(coords))


# ----------------------------------------------------------------------------------------------------------------
# Bouzidi q-map, src/bouzidi_setup.jl:64-167 + src/bouzidi_math.jl:9-102
# ----------------------------------------------------------------------------------------------------------------
def _boundary_cell_lists(mark: np.ndarray):
    """sorted boundary-cell lists (block, then z, y, x) from a bool [8,8,8,nb] (F order) marker"""
    flat = np.flatnonzero(mark.reshape(-1, order="F"))
    b, cell = flat // 512, flat % 512
    return ((b + 1).astype(np.int32), (cell % 8 + 1).astype(np.int8), ((cell // 8) % 8 + 1).astype(np.int8),
            (cell // 64 + 1).astype(np.int8), int(flat.shape[0]))


def compute_bouzidi_qmap_sparse(coords, mesh: SolverMesh, dx: float, offset, shell: Optional[np.ndarray] = None, method: str = "native"):
    """Returns (q_map Float16 [8,8,8,nb,27], cell_block, cell_x, cell_y, cell_z (1-based), n_boundary_cells).

    The reference casts, from every cell of every block that has triangles binned to it (margin 2.5 dx), 26 rays against
    all binned triangles and keeps q = t_min / (dx |c|) if 0 < q <= 1. A link is at most sqrt(3) dx long, so only
    triangles whose bounding box is within sqrt(3) dx of the cell centre can give q <= 1, and a farther triangle can
    never undercut a nearer hit: evaluating exactly those (cell, triangle) pairs gives the same q-map. The list order is
    thread-dependent in the reference and irrelevant (Appendix A.11); here it is sorted.
    method "native": libludwig_setup.so (a thread per block, as the reference); "numpy": the restatement below (the checker)."""
    n = len(coords)
    if method == "native":
        from . import _setup_lib
        lib, c, tri, nt = _native_inputs(coords, mesh, offset)
        q_map = np.zeros((BS, BS, BS, n, 27), dtype=np.float16, order="F")
        mark = np.zeros((BS, BS, BS, n), dtype=np.uint8, order="F")
        nbc = _setup_lib.check(lib.lws_bouzidi_qmap(_ptr(tri), tri.shape[0], float(dx), _ptr(c), n, _ptr(q_map), _ptr(mark), nt),
                               "lws_bouzidi_qmap")
        cb, cx, cy, cz, cnt = _boundary_cell_lists(mark)
        assert cnt == nbc
        return q_map, cb, cx, cy, cz, nbc
    d = _Dense(coords)
    tri = mesh.triangles + np.asarray(offset)
    cell, ti, o = _candidate_pairs(tri.min(axis=1), tri.max(axis=1), math.sqrt(3.0) * dx * 1.0001, dx, d)
    order = np.argsort(cell, kind="stable")
    cell, ti, o = cell[order], ti[order], o[order]
    ucell, first = np.unique(cell, return_index=True)
    v1 = tri[ti, 0]
    e1, e2 = tri[ti, 1] - v1, tri[ti, 2] - v1
    sv = o - v1
    qv = np.cross(sv, e1)
    t_num = (e2 * qv).sum(axis=1)
    EPS = 1e-9
    qdense = np.zeros((27,) + d.shape, dtype=np.float16)
    any_hit = np.zeros(len(ucell), dtype=bool)
    k = -1
    for cz in (-1, 0, 1):
        for cy in (-1, 0, 1):
            for cx in (-1, 0, 1):
                k += 1
                if (cx, cy, cz) == (0, 0, 0):
                    continue
                cm = math.sqrt(cx * cx + cy * cy + cz * cz)
                dn = np.array([cx, cy, cz], dtype=np.float64) / cm
                h = np.cross(dn, e2)
                a = (e1 * h).sum(axis=1)
                good = np.abs(a) >= EPS
                f = 1.0 / np.where(good, a, 1.0)
                u = f * (sv * h).sum(axis=1)
                v = f * (qv * dn).sum(axis=1)
                t = f * t_num
                hit = good & ~((u < 0.0) | (u > 1.0)) & ~((v < 0.0) | (u + v > 1.0)) & (t > EPS)
                tmin = np.minimum.reduceat(np.where(hit, t, np.inf), first)
                q = tmin / (dx * cm)
                ok = np.isfinite(tmin) & (q > 0.0) & (q <= 1.0)
                qdense[k].reshape(-1)[ucell[ok]] = q[ok].astype(np.float16)
                any_hit |= ok
    q_map = np.zeros((BS, BS, BS, n, 27), dtype=np.float16, order="F")
    for kk in range(27):
        q_map[..., kk] = d.to_blocks(qdense[kk], n)
    mark = np.zeros(d.shape, dtype=bool)
    mark.reshape(-1)[ucell[any_hit]] = True
    arr = np.argwhere(d.to_blocks(mark, n))                 # (x, y, z, b) 0-based
    arr = arr[np.lexsort((arr[:, 0], arr[:, 1], arr[:, 2], arr[:, 3]))]
    return (q_map, (arr[:, 3] + 1).astype(np.int32), (arr[:, 0] + 1).astype(np.int8), (arr[:, 1] + 1).astype(np.int8),
            (arr[:, 2] + 1).astype(np.int8), int(arr.shape[0]))


# ----------------------------------------------------------------------------------------------------------------
# orchestration, src/domain.jl:20-266
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class SetupReport:
    """The numbers the reference prints while it builds the domain (its only setup KATs)."""
    level_blocks: List[int] = field(default_factory=list)
    halo_blocks_added: List[Optional[int]] = field(default_factory=list)
    flood_fill_filled: List[int] = field(default_factory=list)
    sponge_fraction: List[float] = field(default_factory=list)
    sponge_max: List[float] = field(default_factory=list)
    near_wall_cells: List[int] = field(default_factory=list)
    bouzidi_cells: List[int] = field(default_factory=list)


def setup_multilevel_domain(cfg: CaseConfig, stl_path: Optional[str] = None, method: Optional[str] = None):
    """setup_multilevel_domain(stl_path) (src/domain.jl:268-280) -> (grids, mesh, params, report).
    method: "native" (default; libludwig_setup.so, threads over blocks) or "numpy" (the restatement the native code is checked
    against; env LUDWIG_SETUP_METHOD). There is no silent fall-back from one to the other."""
    method = method or os.environ.get("LUDWIG_SETUP_METHOD", "native")
    if method not in ("native", "numpy"):
        raise ValueError(f"setup method {method!r}: native or numpy")
    stl = stl_path or os.path.join(cfg.case_dir, cfg.stl_file)
    mesh = load_mesh(stl, scale=cfg.stl_scale)
    params = compute_domain_from_mesh(mesh.min_bounds, mesh.max_bounds, cfg)
    rep = SetupReport()
    off = params.mesh_offset
    pmin, pmax = params.mesh_min + off, params.mesh_max + off
    wake_start_x = pmax[0] - params.reference_length * 0.1
    wake_end_x = pmax[0] + params.reference_length * cfg.wake_length
    wcy, wcz = (pmin[1] + pmax[1]) / 2.0, (pmin[2] + pmax[2]) / 2.0
    ww, wh = (pmax[1] - pmin[1]) * cfg.wake_width_factor, (pmax[2] - pmin[2]) * cfg.wake_height_factor
    wy0, wy1, wz0, wz1 = wcy - ww / 2.0, wcy + ww / 2.0, wcz - wh / 2.0, wcz + wh / 2.0
    grids: List[BlockLevel] = []
    for lvl in range(1, params.num_levels + 1):
        scale = 2 ** (lvl - 1)
        dx = params.dx_coarse / scale
        bmax = (params.bx_max * scale, params.by_max * scale, params.bz_max * scale)
        active: Set[Coord] = set()
        if lvl == 1:
            active = {(bx, by, bz) for bz in range(1, bmax[2] + 1) for by in range(1, bmax[1] + 1) for bx in range(1, bmax[0] + 1)}
        else:
            prev = grids[-1]
            prev_bs = BS * params.dx_coarse / 2 ** (lvl - 2)
            if cfg.refinement_strategy == "geometry_first":
                active |= get_active_blocks_for_level(mesh, dx, off, bmax)
                if cfg.wake_enabled:
                    for (cbx, cby, cbz) in prev.active_block_coords:
                        ox = ((cbx - 1) * prev_bs <= wake_end_x) and (cbx * prev_bs >= wake_start_x)
                        oy = ((cby - 1) * prev_bs <= wy1) and (cby * prev_bs >= wy0)
                        oz = ((cbz - 1) * prev_bs <= wz1) and (cbz * prev_bs >= wz0)
                        if ox and oy and oz:
                            for db in range(8):
                                fb = (2 * cbx - 1 + (db & 1), 2 * cby - 1 + ((db >> 1) & 1), 2 * cbz - 1 + ((db >> 2) & 1))
                                if _inside(fb, bmax):
                                    active.add(fb)
                pset = set(prev.active_block_coords)
                active = {c for c in active if ((c[0] + 1) // 2, (c[1] + 1) // 2, (c[2] + 1) // 2) in pset}
            else:
                for b_idx, (cbx, cby, cbz) in enumerate(prev.active_block_coords):
                    if prev.obstacle[:, :, :, b_idx].any():
                        for db in range(8):
                            active.add((2 * cbx - 1 + (db & 1), 2 * cby - 1 + ((db >> 1) & 1), 2 * cbz - 1 + ((db >> 2) & 1)))
        n_before = len(active)
        add_halo_blocks_with_siblings(active, cfg.refinement_margin, bmax)
        ensure_complete_parent_coverage(active, bmax)
        rep.halo_blocks_added.append(len(active) - n_before if lvl > 1 else None)
        coords = sorted(active)
        table = build_neighbor_table(coords, *bmax)
        obstacle = voxelize_blocks(coords, mesh, dx, off, method=method)
        shell = obstacle.copy()
        rep.flood_fill_filled.append(perform_flood_fill(obstacle, coords, method=method))
        sponge = apply_sponge(coords, params, scale, cfg)
        rep.sponge_fraction.append(float((sponge > 0).mean()))
        rep.sponge_max.append(float(sponge.max()))
        wall = None
        if cfg.wall_model_enabled:
            wall = compute_wall_distances(coords, obstacle, dx, method=method)
            rep.near_wall_cells.append(int((wall != np.float32(100.0)).sum()))
        use_bouzidi = cfg.boundary_method == "bouzidi" and lvl > params.num_levels - cfg.bouzidi_levels
        kw = {}
        if use_bouzidi:
            q, cb, cx, cy, cz, nbc = compute_bouzidi_qmap_sparse(coords, mesh, dx, off, shell, method=method)
            kw = dict(bouzidi_q_map=q, bouzidi_cell_block=cb, bouzidi_cell_x=cx, bouzidi_cell_y=cy, bouzidi_cell_z=cz, n_boundary_cells=nbc)
            rep.bouzidi_cells.append(nbc)
        level = BlockLevel(lvl, coords, table, float(np.float32(dx)), np.float32(1.0) / np.float32(scale), params.tau_levels[lvl - 1],
                           enable_temporal_interpolation=cfg.temporal_interpolation, **kw)
        level.obstacle[...] = obstacle
        level.sponge[...] = sponge
        if wall is not None:
            level.wall_dist[...] = wall
        rep.level_blocks.append(len(coords))
        grids.append(level)
    return grids, mesh, params, rep


def solver_params(cfg: CaseConfig, params: DomainParameters) -> SolverParams:
    """the scalar arguments src/main.jl:176-180 passes to execute_timestep_batch!"""
    return SolverParams(domain_nx=params.nx_coarse, domain_ny=params.ny_coarse, domain_nz=params.nz_coarse,
                        wall_model_active=params.wall_model_active, c_wale=float(cfg.c_wale), nu_sgs_bg=float(cfg.nu_sgs_background),
                        inlet_turbulence=float(cfg.inlet_turbulence_intensity), use_temporal_interp=cfg.temporal_interpolation,
                        sponge_blend_dist=cfg.sponge_blend_distributions, symmetric_analysis=cfg.symmetric_analysis,
                        q_min_threshold=float(cfg.q_min_threshold))
