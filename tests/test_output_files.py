"""Scope row N4: result files. The VTU files are parsed back with a small reader written here (the VTK XML dialect:
inline base64, UInt64 headers, zlib blocks) and compared with the level arrays they were made from."""
import base64
import os
import re
import zlib

import numpy as np
import pytest

from open_ludwig_amd import cases, output

_NP = {"Float32": np.float32, "Float64": np.float64, "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8}


def _b64_len(nbytes):
    return 4 * ((nbytes + 2) // 3)


def read_vtu(path):
    txt = open(path).read()
    compressed = 'compressor="vtkZLibDataCompressor"' in txt
    head = re.search(r'<Piece NumberOfPoints="(\d+)" NumberOfCells="(\d+)">', txt)
    out = {"n_points": int(head.group(1)), "n_cells": int(head.group(2))}
    for m in re.finditer(r'<DataArray type="(\w+)" Name="(\w+)"( NumberOfComponents="(\d+)")? format="binary">([^<]*)</DataArray>', txt):
        dtype, name, ncomp, payload = _NP[m.group(1)], m.group(2), int(m.group(4) or 1), m.group(5)
        if compressed:
            h3 = np.frombuffer(base64.b64decode(payload[:_b64_len(24)]), dtype=np.uint64)
            nblk = int(h3[0])
            hl = _b64_len(8 * (3 + nblk))
            sizes = np.frombuffer(base64.b64decode(payload[:hl]), dtype=np.uint64)[3:]
            blob = base64.b64decode(payload[hl:])
            raw, o = b"", 0
            for s in sizes:
                raw += zlib.decompress(blob[o:o + int(s)])
                o += int(s)
        else:
            n = int(np.frombuffer(base64.b64decode(payload[:_b64_len(8)]), dtype=np.uint64)[0])
            raw = base64.b64decode(payload[_b64_len(8):])[:n]
        a = np.frombuffer(raw, dtype=dtype)
        out[name] = a.reshape(-1, ncomp) if ncomp > 1 else a
    return out


def test_export_block_selection_rule():
    """a parent is dropped only when all 8 children exist (src/io_vtk.jl:27-46)"""
    l1 = [(1, 1, 1), (2, 1, 1)]
    full = [(1 + dx, 1 + dy, 1 + dz) for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)]
    partial = [(3, 1, 1), (4, 1, 1)]
    assert output.select_export_blocks([l1, full + partial]) == [(0, 1)] + [(1, i) for i in range(10)]
    assert output.select_export_blocks([l1, partial]) == [(0, 0), (0, 1), (1, 0), (1, 1)]
    assert output.select_export_blocks([l1]) == [(0, 0), (0, 1)]


@pytest.mark.parametrize("t_step", [3, 4])
def test_flow_vtu_round_trip(tmp_path, t_step):
    grids, params = cases.tunnel_with_sphere((4, 2, 2), levels=2, wall_model=False, temporal=True)
    rng = np.random.default_rng(5)
    for g in grids:
        g.rho[...] = rng.random(g.rho.shape, dtype=np.float32) + 0.5
        g.vel[...] = rng.random(g.vel.shape, dtype=np.float32) - 0.5
        g.vel_temp[...] = rng.random(g.vel.shape, dtype=np.float32) - 0.5
        g.dx = 0.25 / 2 ** (g.level_id - 1)
    grids[0].rho[1, 2, 3, 0] = np.nan
    grids[0].vel[0, 0, 0, 0, 1] = np.inf
    grids[0].vel_temp[0, 0, 0, 0, 1] = -np.inf
    path = output.export_merged_mesh(t_step, grids, lambda lvl, name: getattr(grids[lvl], name), str(tmp_path))
    assert path.endswith("flow_%06d.vtu" % t_step)
    d = read_vtu(path)
    sel = output.select_export_blocks([g.active_block_coords for g in grids])
    assert 0 < len(sel) < sum(g.n_blocks for g in grids), "some level-1 blocks are fully refined and must be dropped"
    assert d["n_cells"] == 512 * len(sel) and d["n_points"] == 729 * len(sel)
    assert (d["types"] == 11).all() and np.array_equal(d["offsets"], 8 * np.arange(1, d["n_cells"] + 1))
    vname = "vel_temp" if t_step % 2 == 0 else "vel"
    for i in (0, len(sel) // 2, len(sel) - 1):
        lvl, b = sel[i]
        g = grids[lvl]
        bc = np.asarray(g.active_block_coords[b])
        for (x, y, z) in ((0, 0, 0), (7, 0, 3), (1, 2, 3), (7, 7, 7)):
            c = i * 512 + x + 8 * y + 64 * z
            pts = d["Points"][d["connectivity"][8 * c: 8 * c + 8]]
            lo = ((bc - 1) * 8 + np.array([x, y, z])).astype(np.float32) * np.float32(g.dx)
            np.testing.assert_allclose(pts.min(axis=0), lo, rtol=1e-6)
            np.testing.assert_allclose(pts.max(axis=0) - pts.min(axis=0), np.float32(g.dx), rtol=1e-5)
            assert np.array_equal(pts[1] - pts[0], [np.float32(pts[1, 0] - pts[0, 0]), 0, 0]) and pts[2, 1] > pts[0, 1] and pts[4, 2] > pts[0, 2]   # voxel order
            want_rho = g.rho[x, y, z, b]
            assert d["Density"][c] == (want_rho if np.isfinite(want_rho) else 0.0)
            want_v = getattr(g, vname)[x, y, z, b, :]
            assert np.array_equal(d["Velocity"][c], np.where(np.isfinite(want_v), want_v, 0.0))
            assert d["Obstacle"][c] == int(g.obstacle[x, y, z, b]) and d["Level"][c] == lvl + 1
    v = d["Velocity"]
    assert np.array_equal(d["VelocityMagnitude"], np.sqrt(v[:, 0] ** 2 + v[:, 1] ** 2 + v[:, 2] ** 2))
    assert np.isfinite(d["Density"]).all() and np.isfinite(v).all()
    assert d["Density"].dtype == np.float32 and d["Obstacle"].dtype == np.uint8 and d["Level"].dtype == np.int32


def test_surface_vtu_and_csv_formats(tmp_path):
    from open_ludwig_amd import preprocess as pp
    from open_ludwig_amd.forces import ForceResult
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    mesh = pp.load_mesh(os.path.join(G, "cube1m.stl"))
    n = mesh.triangles.shape[0]
    rng = np.random.default_rng(2)
    p, sx, sy, sz = (rng.standard_normal(n).astype(np.float32) for _ in range(4))
    p[0] = sx[0] = 0.0
    path = output.save_surface_vtk(str(tmp_path / "surface_000100"), mesh, p, sx, sy, sz)
    d = read_vtu(path)
    assert d["n_cells"] == n and d["n_points"] == 3 * n and (d["types"] == 5).all()
    assert d["Points"].dtype == np.float64 and np.array_equal(d["Points"].reshape(n, 3, 3), mesh.triangles)
    assert np.array_equal(d["Pressure_Pa"], p) and np.array_equal(d["ShearMagnitude_Pa"], np.sqrt(sx ** 2 + sy ** 2 + sz ** 2))
    assert d["MappingQuality"][0] == 0.0 and d["MappingQuality"][1:].all()
    assert np.array_equal(d["Normal"], mesh.normals.astype(np.float32)) and np.array_equal(d["Area_m2"], mesh.areas.astype(np.float32))
    fr = ForceResult(1.5, -2.0e-3, 3.0, 0.1, 0.2, 0.3, 1.0, 0.0, 0.0, 0.5, 0.0, 0.0, 0.471234567, -0.01, 0.002, 0.0, 0.123, 0.0, n)
    row = output.force_csv_row(500, 0.0123456789, fr, np.float32(0.05))
    assert row == "500,1.234568e-02,0.050000,1.500000e+00,-2.000000e-03,3.000000e+00,1.000000e+00,5.000000e-01,1.000000e-01,2.000000e-01,3.000000e-01,0.471235,-0.010000,0.002000,0.123000"
    assert len(row.split(",")) == len(output.FORCE_CSV_HEADER.split(","))
    assert output.walltime_str(3725.5) == "01:02:05.50"
    conv = output.convergence_csv_row(1000, 12.25, 0.5, np.float32(0.05), np.float32(0.9931), 1234.5, 0.0637, None)
    assert conv == "1000,00:00:12.25,0.5,0.05,0.9931,1234.5,0.0637,N/A"
    output.export_surface_loads_csv(str(tmp_path / "loads.csv"), mesh, (1.0, 2.0, 3.0), p, sx, sy, sz)
    lines = open(tmp_path / "loads.csv").read().splitlines()
    assert len(lines) == n + 1 and lines[0].startswith("triangle_id,cx,cy,cz") and lines[1].split(",")[0] == "1"
    assert "Cd = +0.471235" in output.force_summary(fr, 1.225, 4.0, 0.785, 1.0)


def test_parallel_zlib_blocks_give_the_same_bytes(monkeypatch):
    """Arrays above output._PARALLEL_ABOVE bytes have their zlib blocks compressed on a thread pool (the shipped wing's flow file is 9 GB
    of arrays): block boundaries, order and the header are the serial ones, so the encoded text is identical."""
    from open_ludwig_amd import output as out
    rng = np.random.default_rng(5)
    for n in (0, 1, (1 << 15) // 4, (1 << 15) // 4 + 1, 300_001):
        a = (rng.random(n) * 100).astype(np.float32)
        monkeypatch.setattr(out, "_PARALLEL_ABOVE", 1 << 62)
        serial = out._encode(a, True)
        monkeypatch.setattr(out, "_PARALLEL_ABOVE", -1)
        assert out._encode(a, True) == serial, n
