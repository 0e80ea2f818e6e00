"""User-facing outputs of a run (scope row N4): flow VTU, surface VTU, convergence.csv, forces.csv, surface-loads CSV.

Restates src/io_vtk.jl:13-129 (merged multi-level flow mesh), src/forces/io.jl:26-82 (surface VTU), :89-110 (forces.csv),
:166-190 (per-triangle loads CSV) and the convergence.csv lines of src/main.jl:81-82,207-209. What is kept exactly: which
blocks are exported (a block is dropped only when all 8 of its children exist on the next level, io_vtk.jl:27-46), the
order of blocks / points / cells, the voxel connectivity, the velocity buffer choice (`vel_temp` after an even step, `vel`
after an odd one, on EVERY level - io_vtk.jl:56), the NaN/Inf -> 0 scrub, array names and dtypes, and the CSV columns and
printf formats. What is not claimed: byte identity with WriteVTK's files (same VTK XML dialect - inline base64, zlib
blocks, UInt64 headers - but compressor output and attribute order are WriteVTK's own).
"""
from __future__ import annotations

import base64
import os
import zlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .blocks import BLOCK_SIZE

VTK_VOXEL, VTK_TRIANGLE = 11, 5
_ZBLOCK = 1 << 15
_PARALLEL_ABOVE = 64 << 20      # bytes of one array above which its zlib blocks are compressed on all cores


# ----------------------------------------------------------------------------------------------------------------
# VTK XML writer (UnstructuredGrid, inline binary, optional zlib)
# ----------------------------------------------------------------------------------------------------------------
_VTK_TYPE = {np.dtype(np.float32): "Float32", np.dtype(np.float64): "Float64", np.dtype(np.int32): "Int32",
             np.dtype(np.int64): "Int64", np.dtype(np.uint8): "UInt8"}


def _encode(a: np.ndarray, compress: bool) -> str:
    raw = np.ascontiguousarray(a).tobytes()
    if not compress:
        return (base64.b64encode(np.uint64(len(raw)).tobytes()) + base64.b64encode(raw)).decode()
    view = memoryview(raw)
    starts = range(0, len(raw), _ZBLOCK) if len(raw) else [0]
    n_blocks = len(starts)
    if len(raw) > _PARALLEL_ABOVE:   # a shipped-size flow mesh is 9 GB of arrays: zlib releases the GIL, the blocks are independent
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=_zlib_threads()) as pool:
            comp = list(pool.map(lambda i: zlib.compress(view[i:i + _ZBLOCK], 6), starts, chunksize=256))
    else:
        comp = [zlib.compress(view[i:i + _ZBLOCK], 6) for i in starts]
    tail = len(raw) - (n_blocks - 1) * _ZBLOCK
    last = tail if tail != _ZBLOCK else 0
    head = np.array([n_blocks, _ZBLOCK, last] + [len(c) for c in comp], dtype=np.uint64)
    return (base64.b64encode(head.tobytes()) + base64.b64encode(b"".join(comp))).decode()


def _zlib_threads() -> int:
    try:
        return max(1, min(32, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(32, os.cpu_count() or 1))


def _data_array(name: str, a: np.ndarray, compress: bool, ncomp: int = 1) -> str:
    nc = f' NumberOfComponents="{ncomp}"' if ncomp > 1 or name == "Points" else ""
    return f'<DataArray type="{_VTK_TYPE[a.dtype]}" Name="{name}"{nc} format="binary">{_encode(a, compress)}</DataArray>\n'


def write_vtu(filename: str, points: np.ndarray, connectivity: np.ndarray, offsets: np.ndarray, types: np.ndarray,
              cell_data: Sequence[Tuple[str, np.ndarray]], compress: bool = True) -> str:
    """points [n,3]; connectivity 0-based; cell_data: (name, [ncells] or [ncells, ncomp]). Returns the path written."""
    path = filename if filename.endswith(".vtu") else filename + ".vtu"
    comp_attr = ' compressor="vtkZLibDataCompressor"' if compress else ""
    with open(path, "w") as io:
        io.write('<?xml version="1.0" encoding="utf-8"?>\n')
        io.write(f'<VTKFile type="UnstructuredGrid" version="1.0" byte_order="LittleEndian" header_type="UInt64"{comp_attr}>\n')
        io.write("<UnstructuredGrid>\n")
        io.write(f'<Piece NumberOfPoints="{points.shape[0]}" NumberOfCells="{types.shape[0]}">\n')
        io.write("<Points>\n" + _data_array("Points", points, compress, 3) + "</Points>\n")
        io.write("<Cells>\n")
        io.write(_data_array("connectivity", np.asarray(connectivity, dtype=np.int64), compress))       # (no copy when already Int64)
        io.write(_data_array("offsets", np.asarray(offsets, dtype=np.int64), compress))
        io.write(_data_array("types", np.asarray(types, dtype=np.uint8), compress))
        io.write("</Cells>\n<CellData>\n")
        for name, a in cell_data:
            io.write(_data_array(name, a, compress, 1 if a.ndim == 1 else a.shape[1]))
        io.write("</CellData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n")
    return path


# ----------------------------------------------------------------------------------------------------------------
# flow export (src/io_vtk.jl)
# ----------------------------------------------------------------------------------------------------------------
def select_export_blocks(block_coords_per_level: Sequence[Sequence[Tuple[int, int, int]]]) -> List[Tuple[int, int]]:
    """(level index 0-based, block index 0-based) in the reference's order (src/io_vtk.jl:17-46): levels ascending, blocks
    in list order; a block is skipped only when all 8 children (2b-1+d) exist on the next level."""
    sets = [set(map(tuple, np.asarray(c, dtype=np.int64).reshape(-1, 3).tolist())) for c in block_coords_per_level]
    out: List[Tuple[int, int]] = []
    for lvl, coords in enumerate(block_coords_per_level):
        nxt = sets[lvl + 1] if lvl + 1 < len(sets) else None
        for b, (bx, by, bz) in enumerate(np.asarray(coords, dtype=np.int64).reshape(-1, 3).tolist()):
            if nxt is not None:
                kids = sum((2 * bx - 1 + dx, 2 * by - 1 + dy, 2 * bz - 1 + dz) in nxt for dz in (0, 1) for dy in (0, 1) for dx in (0, 1))
                if kids == 8:
                    continue
            out.append((lvl, b))
    return out


def _scrub(a: np.ndarray) -> np.ndarray:
    """replace!(…, NaN32 => 0, Inf32 => 0, -Inf32 => 0) (src/io_vtk.jl:110-111)"""
    a = a.copy()
    a[~np.isfinite(a)] = 0
    return a


def build_flow_mesh(t_step: int, grids, fields) -> Dict[str, np.ndarray]:
    """The arrays of export_merged_mesh_sync before they hit the file. `grids`: host levels (active_block_coords, dx);
    `fields(level_index, name)` returns the named array of that level ('rho', 'vel', 'vel_temp', 'obstacle')."""
    B = BLOCK_SIZE
    valid = select_export_blocks([g.active_block_coords for g in grids])
    n_total = len(valid)
    n_pts, n_cells = (B + 1) ** 3, B ** 3
    vel_name = "vel_temp" if t_step % 2 == 0 else "vel"
    data = {}
    for lvl in sorted({l for l, _ in valid}):
        data[lvl] = (fields(lvl, "rho"), fields(lvl, vel_name), np.asarray(fields(lvl, "obstacle")).astype(bool), np.float32(grids[lvl].dx))
    points = np.empty((n_total, n_pts, 3), dtype=np.float32)
    rho = np.empty((n_total, n_cells), dtype=np.float32)
    vel = np.empty((n_total, n_cells, 3), dtype=np.float32)
    obst = np.empty((n_total, n_cells), dtype=np.uint8)
    level = np.empty((n_total, n_cells), dtype=np.int32)
    pz, py, px = np.meshgrid(np.arange(B + 1), np.arange(B + 1), np.arange(B + 1), indexing="ij")      # px fastest
    pxyz = np.stack([px.reshape(-1), py.reshape(-1), pz.reshape(-1)], axis=1).astype(np.float32)
    lv_of = np.array([l for l, _ in valid], dtype=np.int64)
    blk_of = np.array([b for _, b in valid], dtype=np.int64)

    def cells_of(a, blocks):
        """[8,8,8,nb(,K)] -> [len(blocks), 512(, K)] with the cell index x fastest; whole 512-cell chunks when the array is Fortran-ordered"""
        if a.flags.f_contiguous:
            nb = a.shape[3]
            if a.ndim == 4:
                return np.take(a.T.reshape(nb, n_cells), blocks, axis=0)
            return np.take(a.T.reshape(a.shape[4], nb, n_cells), blocks, axis=1).transpose(1, 2, 0)
        sub = a[:, :, :, blocks]
        if a.ndim == 4:
            return sub.reshape(n_cells, len(blocks), order="F").T
        return sub.reshape(n_cells, len(blocks), a.shape[4], order="F").transpose(1, 0, 2)

    for lvl in sorted(data):                                         # a level at a time (the valid list is level-major, blocks ascending)
        r, v, o, dx = data[lvl]
        rows = np.flatnonzero(lv_of == lvl)
        blocks = blk_of[rows]
        bc = np.asarray(grids[lvl].active_block_coords, dtype=np.int64)[blocks]
        off = ((bc - 1) * B).astype(np.float32)
        points[rows] = (off[:, None, :] + pxyz[None, :, :]) * dx      # (off + p) * dx in Float32, as the reference
        rho[rows] = cells_of(r, blocks)                               # cell order x fastest
        vel[rows] = cells_of(v, blocks)
        obst[rows] = cells_of(o, blocks)
        level[rows] = lvl + 1
    z, y, x = np.meshgrid(np.arange(B), np.arange(B), np.arange(B), indexing="ij")
    sy, sz = B + 1, (B + 1) ** 2
    base = (x + y * sy + z * sz).reshape(-1)
    corner = np.array([0, 1, sy, sy + 1, sz, sz + 1, sz + sy, sz + sy + 1])
    conn_block = base[:, None] + corner[None, :]                      # 0-based within the block's points
    conn = (conn_block[None, :, :] + (np.arange(n_total) * n_pts)[:, None, None]).reshape(-1)
    vel = _scrub(vel.reshape(-1, 3))
    return {
        "points": points.reshape(-1, 3), "connectivity": np.asarray(conn, dtype=np.int64),
        "offsets": np.arange(8, 8 * (n_total * n_cells + 1), 8, dtype=np.int64),
        "types": np.full(n_total * n_cells, VTK_VOXEL, dtype=np.uint8),
        "Density": _scrub(rho.reshape(-1)), "Velocity": vel,
        "VelocityMagnitude": np.sqrt(vel[:, 0] ** 2 + vel[:, 1] ** 2 + vel[:, 2] ** 2),
        "Obstacle": obst.reshape(-1), "Level": level.reshape(-1),
    }


DEFAULT_FLOW_FIELDS = ("Density", "Velocity", "VelocityMagnitude", "Obstacle", "Level")


def export_merged_mesh(t_step: int, grids, fields, out_dir: str, output_fields: Sequence[str] = DEFAULT_FLOW_FIELDS,
                       compress: bool = True) -> Optional[str]:
    """export_merged_mesh_sync (src/io_vtk.jl:13-129): writes <out_dir>/flow_%06d.vtu; None when nothing to export."""
    m = build_flow_mesh(t_step, grids, fields)
    if m["types"].size == 0:
        return None
    cd = [(n, m[n]) for n in DEFAULT_FLOW_FIELDS if n in output_fields]
    return write_vtu(os.path.join(out_dir, "flow_%06d" % t_step), m["points"], m["connectivity"], m["offsets"], m["types"], cd, compress)


# ----------------------------------------------------------------------------------------------------------------
# surface export + CSV (src/forces/io.jl)
# ----------------------------------------------------------------------------------------------------------------
def save_surface_vtk(filename: str, mesh, p, sx, sy, sz) -> str:
    """save_surface_vtk (src/forces/io.jl:26-82): 3 Float64 points per triangle, 8 cell arrays, uncompressed."""
    n = mesh.triangles.shape[0]
    p, sx, sy, sz = (np.asarray(a, dtype=np.float32) for a in (p, sx, sy, sz))
    pts = np.asarray(mesh.triangles, dtype=np.float64).reshape(n * 3, 3)
    quality = ((np.abs(p) > 1e-10) | (np.abs(sx) > 1e-10)).astype(np.float32)
    cd = [("Pressure_Pa", p), ("ShearX_Pa", sx), ("ShearY_Pa", sy), ("ShearZ_Pa", sz),
          ("ShearMagnitude_Pa", np.sqrt(sx ** 2 + sy ** 2 + sz ** 2)),
          ("Normal", np.asarray(mesh.normals, dtype=np.float32)), ("Area_m2", np.asarray(mesh.areas, dtype=np.float32)),
          ("MappingQuality", quality)]
    return write_vtu(filename, pts, np.arange(3 * n, dtype=np.int64), np.arange(1, n + 1, dtype=np.int64) * 3,
                     np.full(n, VTK_TRIANGLE, dtype=np.uint8), cd, compress=False)


FORCE_CSV_HEADER = "Step,Time_s,U_inlet,Fx_N,Fy_N,Fz_N,Fx_p_N,Fx_v_N,Mx_Nm,My_Nm,Mz_Nm,Cd,Cl,Cs,Cmy"
CONVERGENCE_CSV_HEADER = "Step,Walltime,Time_phys_s,U_inlet_lat,Rho_min,MLUPS,Cd,Cl"


def write_force_csv_header(path: str) -> None:
    with open(path, "w") as io:
        io.write(FORCE_CSV_HEADER + "\n")


def force_csv_row(step: int, time_phys: float, fr, u_inlet) -> str:
    """append_force_csv's printf (src/forces/io.jl:99-110)"""
    return ("%d,%.6e,%.6f,%.6e,%.6e,%.6e,%.6e,%.6e,%.6e,%.6e,%.6e,%.6f,%.6f,%.6f,%.6f" %
            (step, time_phys, float(u_inlet), fr.Fx, fr.Fy, fr.Fz, fr.Fx_pressure, fr.Fx_viscous, fr.Mx, fr.My, fr.Mz, fr.Cd, fr.Cl, fr.Cs, fr.Cmy))


def append_force_csv(path: str, step: int, time_phys: float, fr, u_inlet) -> None:
    with open(path, "a") as io:
        io.write(force_csv_row(step, time_phys, fr, u_inlet) + "\n")


def walltime_str(elapsed: float) -> str:
    """src/main.jl:37-40"""
    return "%02d:%02d:%05.2f" % (int(elapsed // 3600), int((elapsed % 3600) // 60), elapsed % 60)


def _shortest(x) -> str:
    """shortest round-trip decimal of a Float32/Float64 (what Julia's string interpolation prints, up to its choice of
    exponent notation for very small / large magnitudes)"""
    return np.format_float_positional(x, unique=True, trim="0") if 1e-5 <= abs(float(x)) < 1e6 or float(x) == 0.0 \
        else np.format_float_scientific(x, unique=True, trim="0", exp_digits=1).replace("e+", "e")


def convergence_csv_row(step: int, elapsed: float, time_phys: float, u_curr, rho_min, mlups: float, cd: Optional[float], cl: Optional[float]) -> str:
    """src/main.jl:207-209: "$diag_step,$(walltime_str()),$time_phys,$u_curr,$(stats.rho_min),$mlups,$cd_str,$cl_str" """
    cd_s = "N/A" if cd is None else "%.4f" % cd
    cl_s = "N/A" if cl is None else "%.4f" % cl
    return ",".join([str(step), walltime_str(elapsed), _shortest(np.float64(time_phys)), _shortest(np.float32(u_curr)),
                     _shortest(np.float32(rho_min)), _shortest(np.float64(mlups)), cd_s, cl_s])


def export_surface_loads_csv(filename: str, mesh, mesh_offset, p, sx, sy, sz) -> None:
    """export_surface_loads_csv (src/forces/io.jl:166-190)"""
    c = np.asarray(mesh.centers, dtype=np.float64) + np.asarray(mesh_offset, dtype=np.float64)[None, :]
    nrm, area = np.asarray(mesh.normals, dtype=np.float64), np.asarray(mesh.areas, dtype=np.float64)
    with open(filename, "w") as io:
        io.write("triangle_id,cx,cy,cz,nx,ny,nz,area_m2,pressure_Pa,shear_x_Pa,shear_y_Pa,shear_z_Pa\n")
        for i in range(c.shape[0]):
            io.write("%d,%.6e,%.6e,%.6e,%.6f,%.6f,%.6f,%.6e,%.6e,%.6e,%.6e,%.6e\n" %
                     (i + 1, c[i, 0], c[i, 1], c[i, 2], nrm[i, 0], nrm[i, 1], nrm[i, 2], area[i], p[i], sx[i], sy[i], sz[i]))


def force_summary(fr, rho_ref: float, u_ref: float, area_ref: float, chord_ref: float) -> str:
    """print_force_summary (src/forces/io.jl:117-160) as a string"""
    q = 0.5 * rho_ref * u_ref ** 2
    lines = ["", "=" * 60, "         AERODYNAMIC FORCES SUMMARY", "=" * 60, "", "Reference Values:",
             "  ρ_ref  = %.4f kg/m³" % rho_ref, "  U_ref  = %.4f m/s" % u_ref, "  A_ref  = %.4f m²" % area_ref,
             "  L_ref  = %.4f m" % chord_ref, "  q_∞    = %.4f Pa" % q, "", "Forces [N]:",
             "  Fx (drag)  = %+.4e  (pressure: %+.4e, viscous: %+.4e)" % (fr.Fx, fr.Fx_pressure, fr.Fx_viscous),
             "  Fy (side)  = %+.4e  (pressure: %+.4e, viscous: %+.4e)" % (fr.Fy, fr.Fy_pressure, fr.Fy_viscous),
             "  Fz (lift)  = %+.4e  (pressure: %+.4e, viscous: %+.4e)" % (fr.Fz, fr.Fz_pressure, fr.Fz_viscous),
             "", "Moments [N·m]:", "  Mx (roll)  = %+.4e" % fr.Mx, "  My (pitch) = %+.4e" % fr.My, "  Mz (yaw)   = %+.4e" % fr.Mz,
             "", "Coefficients:", "  Cd = %+.6f" % fr.Cd, "  Cl = %+.6f" % fr.Cl, "  Cs = %+.6f" % fr.Cs, "  Cmy = %+.6f" % fr.Cmy]
    if abs(fr.Fx) > 1e-10:
        lines += ["", "Drag breakdown: %.1f%% pressure, %.1f%% viscous" % (abs(fr.Fx_pressure) / abs(fr.Fx) * 100, abs(fr.Fx_viscous) / abs(fr.Fx) * 100)]
    lines += ["=" * 60, ""]
    return "\n".join(lines)
