"""Surface forces (scope row N2): Cd / Cl / Cs / Cm from rho, u of the finest level - restates src/forces/surface.jl.

  compute_stress_from_cell   src/forces/surface.jl:32-96
  map_stresses_kernel!       src/forces/surface.jl:138-266   (nearest fluid cell in expanding shells, radius <= 5)
  integrate_forces_kernel!   src/forces/surface.jl:282-366   (9 Float32 sums)
  integrate_surface_forces!  src/forces/surface.jl:467-571   (symmetry doubling, coefficients)

This is a diagnostic that runs every `diag_freq` steps on <= 63k triangles; it is host numpy on fields downloaded from
the device (14 MB for the 0.88 M-cell finest level of ball1m), Float32 arithmetic like the reference kernels.
The reference sums with 9 contended Float32 atomics, i.e. in an unspecified order; here the sums are numpy Float32
pairwise sums (deterministic), which is one of the orders the reference could have taken up to rounding.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional
from typing import Tuple

import numpy as np

from .blocks import BLOCK_SIZE

f32 = np.float32


@dataclass
class ForceResult:
    Fx: float; Fy: float; Fz: float
    Mx: float; My: float; Mz: float
    Fx_pressure: float; Fy_pressure: float; Fz_pressure: float
    Fx_viscous: float; Fy_viscous: float; Fz_viscous: float
    Cd: float; Cl: float; Cs: float; Cmx: float; Cmy: float; Cmz: float
    coverage: int
    maps: Optional[tuple] = None      # (p, tau_x, tau_y, tau_z) per triangle [Pa], kept for the surface VTU / loads CSV


@dataclass
class NearestCells:
    """Per triangle: the fluid cell map_stresses_kernel! ends up reading (src/forces/surface.jl:191-240). It depends on the
    geometry only (obstacle mask, block layout, triangle centres), never on the flow."""
    found: np.ndarray       # bool [n_tri]
    block: np.ndarray       # int64 [n_tri] 0-based block id (0 where not found)
    lx: np.ndarray; ly: np.ndarray; lz: np.ndarray     # int64 [n_tri] 0-based local coords
    wall_dist: np.ndarray   # float32 [n_tri] sqrt(d2) / dx of the winning cell (0.5 where not found)


def nearest_fluid_cells(mesh, obstacle, block_pointer, dx, params, search_radius: int = 5) -> NearestCells:
    """The search of map_stresses_kernel! for all triangles at once: shells of radius 0..search_radius around the cell
    holding the triangle centre, scan order dz, dy, dx, the first strictly smaller distance wins, the search stops after a
    shell of radius > 1 once a cell was found."""
    B = BLOCK_SIZE
    n = mesh.centers.shape[0]
    dxf = f32(dx)
    off = params.mesh_offset.astype(np.float32)
    c = mesh.centers.astype(np.float32)
    tx, ty, tz = c[:, 0] + off[0], c[:, 1] + off[1], c[:, 2] + off[2]
    g_x = np.floor(tx / dxf).astype(np.int32) + 1
    g_y = np.floor(ty / dxf).astype(np.int32) + 1
    g_z = np.floor(tz / dxf).astype(np.int32) + 1
    dimx, dimy, dimz = block_pointer.shape
    best_d = np.full(n, f32(1e10), dtype=np.float32)
    best_b = np.zeros(n, dtype=np.int64)
    best_l = np.zeros((n, 3), dtype=np.int64)
    best_wd = np.full(n, f32(0.5), dtype=np.float32)
    found = np.zeros(n, dtype=bool)
    bp = np.asarray(block_pointer)
    for radius in range(0, search_radius + 1):
        active = ~(found & (radius > 1))                    # "if found_fluid && radius > 1: break", evaluated per shell
        if not active.any():
            break
        for dz in range(-radius, radius + 1):
            for dy in range(-radius, radius + 1):
                for ddx in range(-radius, radius + 1):
                    if radius > 0 and not (abs(ddx) == radius or abs(dy) == radius or abs(dz) == radius):
                        continue
                    cgx, cgy, cgz = g_x + ddx, g_y + dy, g_z + dz
                    ok = active & (cgx >= 1) & (cgy >= 1) & (cgz >= 1)
                    bx, by, bz = (cgx - 1) // B + 1, (cgy - 1) // B + 1, (cgz - 1) // B + 1
                    ok &= (bx >= 1) & (bx <= dimx) & (by >= 1) & (by <= dimy) & (bz >= 1) & (bz <= dimz)
                    bidx = np.zeros(n, dtype=np.int64)
                    bidx[ok] = bp[bx[ok] - 1, by[ok] - 1, bz[ok] - 1]
                    ok &= bidx > 0
                    if not ok.any():
                        continue
                    lx, ly, lz = (cgx - 1) % B, (cgy - 1) % B, (cgz - 1) % B
                    idx = np.flatnonzero(ok)
                    fluid = ~obstacle[lx[idx], ly[idx], lz[idx], bidx[idx] - 1]
                    idx = idx[fluid]
                    if idx.size == 0:
                        continue
                    ccx = (cgx[idx].astype(np.float32) - f32(0.5)) * dxf
                    ccy = (cgy[idx].astype(np.float32) - f32(0.5)) * dxf
                    ccz = (cgz[idx].astype(np.float32) - f32(0.5)) * dxf
                    d2 = (tx[idx] - ccx) ** 2 + (ty[idx] - ccy) ** 2 + (tz[idx] - ccz) ** 2
                    better = d2 < best_d[idx]
                    j = idx[better]
                    if j.size == 0:
                        continue
                    best_d[j] = d2[better]
                    best_b[j] = bidx[j] - 1
                    best_l[j, 0] = lx[j]; best_l[j, 1] = ly[j]; best_l[j, 2] = lz[j]
                    best_wd[j] = np.sqrt(d2[better]) / dxf
                    found[j] = True
    return NearestCells(found, best_b, best_l[:, 0], best_l[:, 1], best_l[:, 2], best_wd)


def stress_from_cells(best_rho, best_u, best_wd, found, normals, tau, params):
    """compute_stress_from_cell (src/forces/surface.jl:32-96) for arrays of winning cells: rho [n], u [n,3], wall distance
    [n] (lattice units), found [n], triangle normals [n,3]. Returns (p, tau_x, tau_y, tau_z) Float32 [Pa]."""
    nrm = normals.astype(np.float32)
    pressure_scale = f32(params.rho_physical * params.velocity_scale * params.velocity_scale)
    stress_scale = pressure_scale
    wall_dist = np.maximum(best_wd, f32(0.5))
    p_phys = ((best_rho - f32(1.0)) / f32(3.0)) * pressure_scale
    ux, uy, uz = best_u[:, 0], best_u[:, 1], best_u[:, 2]
    udn = ux * nrm[:, 0] + uy * nrm[:, 1] + uz * nrm[:, 2]
    utx, uty, utz = ux - udn * nrm[:, 0], uy - udn * nrm[:, 1], uz - udn * nrm[:, 2]
    umag = np.sqrt(utx * utx + uty * uty + utz * utz)
    nu_lat = (f32(tau) - f32(0.5)) / f32(3.0)
    use = (umag > f32(1e-10)) & (wall_dist > f32(0.01))
    safe = np.where(use, umag, f32(1.0))
    tmag = (best_rho * nu_lat * umag / wall_dist) * stress_scale
    tau_x = np.where(use, (utx / safe) * tmag, f32(0.0)).astype(np.float32)
    tau_y = np.where(use, (uty / safe) * tmag, f32(0.0)).astype(np.float32)
    tau_z = np.where(use, (utz / safe) * tmag, f32(0.0)).astype(np.float32)
    p = np.where(found, p_phys, f32(0.0)).astype(np.float32)
    tau_x = np.where(found, tau_x, f32(0.0)); tau_y = np.where(found, tau_y, f32(0.0)); tau_z = np.where(found, tau_z, f32(0.0))
    return p, tau_x.astype(np.float32), tau_y.astype(np.float32), tau_z.astype(np.float32)


def map_surface_stresses(mesh, rho, vel, obstacle, block_pointer, dx, tau, params, search_radius: int = 5):
    """map_stresses_kernel! for all triangles at once. rho [8,8,8,nb], vel [8,8,8,nb,3], obstacle bool, block_pointer
    [dimx,dimy,dimz] 1-based. Returns (p, tau_x, tau_y, tau_z) Float32 per triangle."""
    nc = nearest_fluid_cells(mesh, obstacle, block_pointer, dx, params, search_radius)
    n = mesh.centers.shape[0]
    best_rho = np.ones(n, dtype=np.float32)
    best_u = np.zeros((n, 3), dtype=np.float32)
    j = np.flatnonzero(nc.found)
    best_rho[j] = rho[nc.lx[j], nc.ly[j], nc.lz[j], nc.block[j]]
    for comp in range(3):
        best_u[j, comp] = vel[nc.lx[j], nc.ly[j], nc.lz[j], nc.block[j], comp]
    return stress_from_cells(best_rho, best_u, nc.wall_dist, nc.found, mesh.normals, tau, params)


def map_surface_stresses_device(mesh, device_level, dx, tau, params, search_radius: int = 5, vel_name: str = "vel"):
    """map_stresses_kernel! on the device that holds the level (ludwig_map_surface_stresses): same search, same Float32
    expressions as map_surface_stresses above - the two are tested to agree bit for bit - without moving rho / vel to the host."""
    import ctypes as C
    from . import _lib
    n = mesh.centers.shape[0]
    centers = np.ascontiguousarray(mesh.centers, dtype=np.float32)
    normals = np.ascontiguousarray(mesh.normals, dtype=np.float32)
    off = params.mesh_offset.astype(np.float32)
    scale = f32(params.rho_physical * params.velocity_scale * params.velocity_scale)
    sp = _lib.SurfaceParams(float(f32(dx)), float(f32(tau)), float(off[0]), float(off[1]), float(off[2]), float(scale), float(scale), int(search_radius))
    out = [np.empty(n, dtype=np.float32) for _ in range(4)]
    _lib.check(_lib.load().ludwig_map_surface_stresses(device_level.handle, _lib.FIELD_NAMES[vel_name], n, centers.ctypes.data, normals.ctypes.data,
                                                       C.byref(sp), *[a.ctypes.data for a in out]))
    return tuple(out)


def partial_force_sums(mesh, p, tau_x, tau_y, tau_z, params, select=None) -> np.ndarray:
    """The nine Float32 sums of integrate_forces_kernel! (src/forces/surface.jl:282-366) - pressure force, viscous force,
    moment about params.moment_center - over all triangles, or over the triangles `select` (an index array: one rank's
    share in a distributed run). Returns Float32 [9] = Fp(3), Fv(3), M(3); `coverage` count is returned separately."""
    off = params.mesh_offset.astype(np.float32)
    c = mesh.centers.astype(np.float32)
    nrm = mesh.normals.astype(np.float32)
    A = mesh.areas.astype(np.float32)
    if select is not None:
        c, nrm, A = c[select], nrm[select], A[select]
    mc = np.asarray(params.moment_center, dtype=np.float32)
    cx, cy, cz = c[:, 0] + off[0], c[:, 1] + off[1], c[:, 2] + off[2]
    dFp = np.stack([-p * nrm[:, 0] * A, -p * nrm[:, 1] * A, -p * nrm[:, 2] * A], axis=1)
    dFv = np.stack([tau_x * A, tau_y * A, tau_z * A], axis=1)
    dF = dFp + dFv
    rx, ry, rz = cx - mc[0], cy - mc[1], cz - mc[2]
    dM = np.stack([ry * dF[:, 2] - rz * dF[:, 1], rz * dF[:, 0] - rx * dF[:, 2], rx * dF[:, 1] - ry * dF[:, 0]], axis=1)
    out = np.zeros(9, dtype=np.float32)
    for i in range(3):
        out[i] = np.sum(dFp[:, i], dtype=np.float32)
        out[3 + i] = np.sum(dFv[:, i], dtype=np.float32)
        out[6 + i] = np.sum(dM[:, i], dtype=np.float32)
    return out


def finish_forces(sums, coverage: int, params, symmetric: bool = False) -> ForceResult:
    """integrate_surface_forces! after the kernel (src/forces/surface.jl:507-571): symmetry doubling, coefficients."""
    Fp = [float(sums[i]) for i in range(3)]
    Fv = [float(sums[3 + i]) for i in range(3)]
    M = [float(sums[6 + i]) for i in range(3)]
    if symmetric:
        Fp[0] *= 2.0; Fp[2] *= 2.0; Fv[0] *= 2.0; Fv[2] *= 2.0
        M[1] *= 2.0
        Fp[1] = 0.0; Fv[1] = 0.0; M[0] = 0.0; M[2] = 0.0
    F = [Fp[i] + Fv[i] for i in range(3)]
    q_inf = 0.5 * params.rho_physical * params.u_physical ** 2
    F_ref = q_inf * params.reference_area
    M_ref = F_ref * params.reference_chord
    cd = cl = cs = cmx = cmy = cmz = 0.0
    if F_ref > 1e-10:
        cd, cl, cs = F[0] / F_ref, F[2] / F_ref, F[1] / F_ref
    if M_ref > 1e-10:
        cmx, cmy, cmz = M[0] / M_ref, M[1] / M_ref, M[2] / M_ref
    return ForceResult(F[0], F[1], F[2], M[0], M[1], M[2], Fp[0], Fp[1], Fp[2], Fv[0], Fv[1], Fv[2], cd, cl, cs, cmx, cmy, cmz,
                       int(coverage))


def combine_partial_sums(part: np.ndarray, comm_device=None):
    """Distributed integrate_forces_kernel!: `part` = this rank's Float32 [10] (nine sums + coverage count). One all-gather
    of 10 floats per rank over the default torch.distributed group, then every rank adds the rows in rank order in Float32 -
    deterministic, the same on every rank, independent of the collective's internal order. Returns (sums [9], coverage)."""
    import torch
    import torch.distributed as dist
    mine = torch.as_tensor(np.asarray(part, dtype=np.float32), device=comm_device if comm_device is not None else torch.device("cpu"))
    rows = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, mine)
    total = np.zeros(9, dtype=np.float32)
    cov = 0
    for a in rows:
        a = a.cpu().numpy()
        total = (total + a[:9]).astype(np.float32)
        cov += int(a[9])
    return total, cov


def integrate_surface_forces(mesh, p, tau_x, tau_y, tau_z, params, symmetric: bool = False) -> ForceResult:
    sums = partial_force_sums(mesh, p, tau_x, tau_y, tau_z, params)
    return finish_forces(sums, int(np.count_nonzero(np.abs(p) > 1e-10)), params, symmetric)


def compute_aerodynamics(mesh, level_host, rho, vel, params, symmetric: bool = False, search_radius: int = 5) -> ForceResult:
    """compute_aerodynamics! (src/forces/surface.jl:592-600) on the finest level. `vel` must be the level's `vel` buffer
    (not vel_temp), as the reference reads it (src/forces/surface.jl:412, Appendix A.13)."""
    p, tx, ty, tz = map_surface_stresses(mesh, rho, vel, level_host.obstacle, level_host.block_pointer, level_host.dx, level_host.tau,
                                         params, search_radius)
    fr = integrate_surface_forces(mesh, p, tx, ty, tz, params, symmetric)
    fr.maps = (p, tx, ty, tz)
    return fr
