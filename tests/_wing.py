"""A synthetic half wing for the symmetric-analysis tests: tapered, swept, NACA 0012 sections at 5 degrees of incidence,
root on the symmetry plane y = 0. Proportions follow the reference's Wing_5_deg case (chord 5.34 m at the root, 14 m
reference length); the geometry itself is generated here (binary STL), not taken from the reference."""
import struct

import numpy as np


def naca0012(x):
    t = 0.12
    return 5 * t * (0.2969 * np.sqrt(x) - 0.1260 * x - 0.3516 * x ** 2 + 0.2843 * x ** 3 - 0.1036 * x ** 4)


def half_wing_triangles(span=7.0, root_chord=5.34, tip_chord=2.2, sweep=2.5, incidence_deg=5.0, n_around=24, n_span=14):
    xs = 0.5 * (1 - np.cos(np.linspace(0.0, np.pi, n_around + 1)))          # 0..1, clustered at both ends
    upper = np.stack([xs, naca0012(xs)], 1)
    lower = np.stack([xs[-2:0:-1], -naca0012(xs[-2:0:-1])], 1)
    prof = np.concatenate([upper, lower])                                    # closed loop, 2 * n_around points
    a = np.deg2rad(incidence_deg)
    rings = []
    for j in range(n_span + 1):
        s = j / n_span
        chord = root_chord + (tip_chord - root_chord) * s
        x = sweep * s + prof[:, 0] * chord
        z = prof[:, 1] * chord
        xr = x * np.cos(a) + z * np.sin(a)
        zr = -x * np.sin(a) + z * np.cos(a)
        rings.append(np.stack([xr, np.full_like(xr, span * s), zr], 1))
    tris = []
    m = prof.shape[0]
    for j in range(n_span):
        r0, r1 = rings[j], rings[j + 1]
        for i in range(m):
            i2 = (i + 1) % m
            tris.append((r0[i], r1[i], r0[i2]))
            tris.append((r0[i2], r1[i], r1[i2]))
    for ring, flip in ((rings[0], False), (rings[-1], True)):                 # caps (fans around the section centre)
        c = ring.mean(axis=0)
        for i in range(m):
            i2 = (i + 1) % m
            tris.append((c, ring[i], ring[i2]) if flip else (c, ring[i2], ring[i]))
    return np.asarray(tris, dtype=np.float32)


def write_binary_stl(path, tris):
    tris = np.asarray(tris, dtype=np.float32)
    n = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    with open(path, "wb") as f:
        f.write(b"synthetic half wing".ljust(80, b" "))
        f.write(struct.pack("<I", tris.shape[0]))
        for i in range(tris.shape[0]):
            f.write(struct.pack("<12fH", *n[i], *tris[i].reshape(-1), 0))


# Wing_5_deg parameters at a size the CPU oracle steps in seconds: 3 levels, dx_fine = 14 m / 112, a short ramp
TEST_OVERRIDES = {"basic": {"surface_resolution": 112, "num_levels": 3, "simulation": {"ramp_steps": 40}}}
