"""Block data model: host mirror of the reference's BlockLevel and its device counterpart.

Reference: src/blocks.jl (struct :16-65, constructor :89-188, copy_to_old! :199-205, has_temporal_storage :208),
src/domain_topology.jl:135-160 (build_neighbor_table), src/physics_v2.jl:99-117 (lattice tables).

Host arrays are numpy arrays in Fortran (= Julia column-major) order with the reference's shapes, so
`level.f[x-1, y-1, z-1, b-1, k-1]` is the reference's `level.f[x, y, z, b, k]` and the raw memory is
byte-compatible with what libludwig_hip.so expects. Index tables keep the reference's 1-based values (0 = absent).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

BLOCK_SIZE = 8  # src/blocks.jl:14


def build_lattice_arrays():
    """build_lattice_arrays_gpu (src/physics_v2.jl:99-117): (cx, cy, cz, w, opp, mirror_y, mirror_z), 1-based tables."""
    cx, cy, cz, w = [], [], [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                cx.append(dx); cy.append(dy); cz.append(dz)
                d2 = dx * dx + dy * dy + dz * dz
                w.append(np.float32(8) / np.float32(27) if d2 == 0 else np.float32(2) / np.float32(27) if d2 == 1
                         else np.float32(1) / np.float32(54) if d2 == 2 else np.float32(1) / np.float32(216))
    cx, cy, cz = (np.array(a, dtype=np.int32) for a in (cx, cy, cz))
    w = np.array(w, dtype=np.float32)
    opp = np.zeros(27, np.int32); my = np.zeros(27, np.int32); mz = np.zeros(27, np.int32)
    for i in range(27):
        for j in range(27):
            if cx[j] == -cx[i] and cy[j] == -cy[i] and cz[j] == -cz[i]: opp[i] = j + 1
            if cx[j] == cx[i] and cy[j] == -cy[i] and cz[j] == cz[i]: my[i] = j + 1
            if cx[j] == cx[i] and cy[j] == cy[i] and cz[j] == -cz[i]: mz[i] = j + 1
    return cx, cy, cz, w, opp, my, mz


def build_neighbor_table(active_coords: Sequence[Tuple[int, int, int]], bx_max: int, by_max: int, bz_max: int,
                         periodic: Tuple[bool, bool, bool] = (False, False, False)) -> np.ndarray:
    """src/domain_topology.jl:135-160. Returns Int32 [n_blocks, 27] (Fortran order), 1-based, 0 = none.

    `periodic` is this package's extension: the reference has no periodic boundary, but its kernel treats
    any non-zero entry as an interior neighbour, so a wrapped table yields a periodic box without a kernel change.
    """
    coords = np.asarray(active_coords, dtype=np.int64).reshape(-1, 3)
    n = coords.shape[0]
    table = np.zeros((n, 27), dtype=np.int32, order="F")
    if n == 0:
        return table
    ptr = np.zeros((bx_max, by_max, bz_max), dtype=np.int32)
    ok = ((coords[:, 0] >= 1) & (coords[:, 0] <= bx_max) & (coords[:, 1] >= 1) & (coords[:, 1] <= by_max)
          & (coords[:, 2] >= 1) & (coords[:, 2] <= bz_max))
    ptr[coords[ok, 0] - 1, coords[ok, 1] - 1, coords[ok, 2] - 1] = (np.nonzero(ok)[0] + 1).astype(np.int32)
    dims = np.array([bx_max, by_max, bz_max])
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                d = (dx + 1) + (dy + 1) * 3 + (dz + 1) * 9
                nb = coords + np.array([dx, dy, dz])
                inside = np.ones(n, dtype=bool)
                for a in range(3):
                    if periodic[a]:
                        nb[:, a] = (nb[:, a] - 1) % dims[a] + 1
                    else:
                        inside &= (nb[:, a] >= 1) & (nb[:, a] <= dims[a])
                vals = np.zeros(n, dtype=np.int32)
                vals[inside] = ptr[nb[inside, 0] - 1, nb[inside, 1] - 1, nb[inside, 2] - 1]
                table[:, d] = vals
    return table


def _f(shape, fill=0.0, dtype=np.float32):
    if fill == 0.0:
        return np.zeros(shape, dtype=dtype, order="F")      # untouched pages: a 256^3 level is several GB of these
    a = np.empty(shape, dtype=dtype, order="F")
    a[...] = fill
    return a


class BlockLevel:
    """Host mirror of the reference's BlockLevel (src/blocks.jl:16-65, constructor :89-188). Same field names."""

    def __init__(self, level_id: int, active_coords: Sequence[Tuple[int, int, int]], neighbor_table: np.ndarray,
                 dx: float, dt: float, tau: float, *, bouzidi_q_map=None, bouzidi_cell_block=None,
                 bouzidi_cell_x=None, bouzidi_cell_y=None, bouzidi_cell_z=None, bouzidi_tri_map=None,
                 n_boundary_cells: int = 0, enable_temporal_interpolation: bool = True):
        B = BLOCK_SIZE
        self.level_id = int(level_id)
        self.dx = float(dx)
        self.dt = np.float32(dt)
        self.tau = np.float32(tau)
        self.active_block_coords: List[Tuple[int, int, int]] = [tuple(int(v) for v in c) for c in active_coords]
        n = len(self.active_block_coords)
        if n:
            arr = np.asarray(self.active_block_coords)
            bx_max, by_max, bz_max = (int(arr[:, i].max()) for i in range(3))
        else:
            bx_max = by_max = bz_max = 0
        self.grid_dim_x, self.grid_dim_y, self.grid_dim_z = bx_max, by_max, bz_max
        self.block_pointer = np.zeros((bx_max, by_max, bz_max), dtype=np.int32, order="F")
        for i, (bx, by, bz) in enumerate(self.active_block_coords):
            self.block_pointer[bx - 1, by - 1, bz - 1] = i + 1

        self.rho = _f((B, B, B, n), 1.0)
        self.vel = _f((B, B, B, n, 3))
        self.vel_temp = _f((B, B, B, n, 3))
        temporal = enable_temporal_interpolation and n > 0
        self.rho_old = _f((B, B, B, n), 1.0) if temporal else _f((1, 1, 1, 1), 1.0)
        self.vel_old = _f((B, B, B, n, 3)) if temporal else _f((1, 1, 1, 1, 3))
        self.f = _f((B, B, B, n, 27))
        self.f_temp = _f((B, B, B, n, 27))
        self.f_post_collision = _f((B, B, B, n, 27)) if n_boundary_cells > 0 else _f((1, 1, 1, 1, 27))
        self.f_old = _f((B, B, B, n, 27)) if temporal else _f((1, 1, 1, 1, 27))
        self.wall_dist = _f((B, B, B, n), 100.0)
        self.obstacle = np.zeros((B, B, B, n), dtype=np.bool_, order="F")
        self.sponge = _f((B, B, B, n))

        self.neighbor_table = np.asfortranarray(neighbor_table, dtype=np.int32)
        assert self.neighbor_table.shape == (n, 27)
        self.map_x = np.array([c[0] for c in self.active_block_coords], dtype=np.int32)
        self.map_y = np.array([c[1] for c in self.active_block_coords], dtype=np.int32)
        self.map_z = np.array([c[2] for c in self.active_block_coords], dtype=np.int32)

        self.bouzidi_enabled = n_boundary_cells > 0 and bouzidi_q_map is not None
        if self.bouzidi_enabled:
            self.bouzidi_q_map = np.asfortranarray(bouzidi_q_map, dtype=np.float16)
            self.bouzidi_cell_block = np.asarray(bouzidi_cell_block, dtype=np.int32)
            self.bouzidi_cell_x = np.asarray(bouzidi_cell_x, dtype=np.int8)
            self.bouzidi_cell_y = np.asarray(bouzidi_cell_y, dtype=np.int8)
            self.bouzidi_cell_z = np.asarray(bouzidi_cell_z, dtype=np.int8)
            self.bouzidi_tri_map = (np.asfortranarray(bouzidi_tri_map, dtype=np.int32) if bouzidi_tri_map is not None
                                    else np.zeros((1, 1, 1, 1, 27), np.int32, order="F"))
        else:
            self.bouzidi_q_map = np.zeros((1, 1, 1, 1, 27), np.float16, order="F")
            self.bouzidi_cell_block = np.zeros(0, np.int32)
            self.bouzidi_cell_x = np.zeros(0, np.int8)
            self.bouzidi_cell_y = np.zeros(0, np.int8)
            self.bouzidi_cell_z = np.zeros(0, np.int8)
            self.bouzidi_tri_map = np.zeros((1, 1, 1, 1, 27), np.int32, order="F")
        self.n_boundary_cells = int(n_boundary_cells)
        # multi-GPU extension (no reference counterpart): blocks [n_owned, n_blocks) are ghosts
        self.n_owned = n
        self.comm_boundary: Optional[np.ndarray] = None

    @property
    def n_blocks(self) -> int:
        return len(self.active_block_coords)


def has_temporal_storage(level) -> bool:
    """src/blocks.jl:208"""
    if isinstance(level, DeviceLevel):
        return level.has_temporal_storage
    return level.f_old.size > 27


def copy_to_old(level, f_current_name: str, vel_current_name: str, t_sub: Optional[int] = None) -> None:
    """copy_to_old!(level, f_current, vel_current) (src/blocks.jl:199-205) for a host BlockLevel."""
    if level.f_old.size > 27:
        level.f_old[...] = getattr(level, f_current_name)
        level.rho_old[...] = level.rho
        level.vel_old[...] = getattr(level, vel_current_name)


_FIELD_DTYPES = {"obstacle": np.uint8}


class DeviceLevel:
    """`adapt(backend, level)` (src/blocks.jl:67-87): a BlockLevel whose arrays live in MI355X HBM.

    Owns an opaque LudwigLevel handle. Field access goes through download()/upload() (`Array(level.f)` /
    `copyto!(level.f, host)` in the reference); there is no host shadow copy.
    """

    def __init__(self, host: BlockLevel, device: int = 0, upload_state: bool = True):
        """upload_state=False: skip copying the host level's f / rho / vel arrays (the caller initialises the state on the device,
        e.g. `init_equilibrium`, src/main.jl:109-134) - on a 55 M-cell level that is 25 GB of zeros over PCIe"""
        lib = _lib.load()
        self._lib = lib
        self.level_id = host.level_id
        self.tau = np.float32(host.tau)
        self.n_blocks = host.n_blocks
        self.n_owned = int(host.n_owned)
        self.active_block_coords = host.active_block_coords
        self.grid_dim_x, self.grid_dim_y, self.grid_dim_z = host.grid_dim_x, host.grid_dim_y, host.grid_dim_z
        self.bouzidi_enabled = bool(host.bouzidi_enabled)
        self.n_boundary_cells = host.n_boundary_cells if host.bouzidi_enabled else 0
        self.device = device
        keep = []   # keep converted arrays alive across the call

        def ptr(a, dtype=None):
            if a is None:
                return None
            a = np.asfortranarray(a, dtype=dtype) if dtype is not None else np.asfortranarray(a)
            keep.append(a)
            return a.ctypes.data

        h = _lib.LevelHost()
        h.level_id = host.level_id
        h.n_blocks = host.n_blocks
        # ABI: 0 = all blocks owned (single device), -1 = none (a level this rank only holds ghost copies of)
        h.n_owned = (int(host.n_owned) if host.n_owned > 0 else -1) if host.n_owned != host.n_blocks else 0
        h.tau = float(host.tau)
        h.grid_dim_x, h.grid_dim_y, h.grid_dim_z = host.grid_dim_x, host.grid_dim_y, host.grid_dim_z
        h.block_pointer = ptr(host.block_pointer, np.int32) if host.block_pointer.size else None
        h.neighbor_table = ptr(host.neighbor_table, np.int32)
        h.map_x, h.map_y, h.map_z = ptr(host.map_x, np.int32), ptr(host.map_y, np.int32), ptr(host.map_z, np.int32)
        h.obstacle = ptr(host.obstacle.view(np.uint8)) if host.obstacle.any() else None
        h.sponge = ptr(host.sponge, np.float32) if (host.sponge > 0).any() else None
        h.wall_dist = ptr(host.wall_dist, np.float32) if (host.wall_dist != 100.0).any() else None
        h.enable_temporal_interpolation = 1 if host.f_old.size > 27 else 0
        h.n_boundary_cells = host.n_boundary_cells
        # multi-GPU, Bouzidi level: `post_collision_readers` (element offsets into f_post_collision that a PEER reads - this rank's
        # f_post send lists) narrows the store to the rows with a reader; without it `force_post_collision` stores every block
        readers = getattr(host, "post_collision_readers", None)
        if (getattr(host, "force_post_collision", False) or readers is not None) and host.n_boundary_cells == 0:
            h.n_boundary_cells = -1        # multi-GPU: store f_post_collision although no Bouzidi cell is owned here
        if host.bouzidi_enabled:
            h.bouzidi_q_map = ptr(host.bouzidi_q_map.view(np.uint16))
            h.bouzidi_cell_block = ptr(host.bouzidi_cell_block, np.int32)
            h.bouzidi_cell_x = ptr(host.bouzidi_cell_x, np.int8)
            h.bouzidi_cell_y = ptr(host.bouzidi_cell_y, np.int8)
            h.bouzidi_cell_z = ptr(host.bouzidi_cell_z, np.int8)
        if host.comm_boundary is not None:
            h.comm_boundary = ptr(host.comm_boundary, np.uint8)
        h.store_post_collision_everywhere = 1 if (getattr(host, "force_post_collision", False) and readers is None) else 0
        handle = C.c_void_p()
        _lib.check(lib.ludwig_level_create(C.byref(h), device, C.byref(handle)))
        self._h = handle
        if readers is not None and self.n_blocks > 0:
            self.add_post_collision_readers(readers)
        # state: the constructor defaults already match; copy whatever the host level holds
        if self.n_blocks == 0 or not upload_state:
            return
        for name in ("f", "f_temp", "rho", "vel", "vel_temp"):
            self.upload(name, getattr(host, name))
        if self.has_temporal_storage:
            for name in ("f_old", "rho_old", "vel_old"):
                self.upload(name, getattr(host, name))
        if host.n_boundary_cells > 0:
            self.upload("f_post_collision", host.f_post_collision)

    # -- life cycle --
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.ludwig_level_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("DeviceLevel is closed")
        return self._h

    def info(self) -> _lib.LevelInfo:
        info = _lib.LevelInfo()
        _lib.check(self._lib.ludwig_level_info(self.handle, C.byref(info)))
        return info

    @property
    def has_temporal_storage(self) -> bool:
        return bool(self.info().has_temporal_storage)

    @property
    def has_post_collision(self) -> bool:
        return bool(self.info().has_post_collision)

    def add_post_collision_readers(self, offsets: np.ndarray) -> None:
        """f_post_collision elements read from outside this level's own cell list (a peer's links across a cut): their x-rows join the store"""
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        _lib.check(self._lib.ludwig_level_add_post_collision_readers(self.handle, o.ctypes.data if o.size else None, o.size))

    def set_stream(self, hip_stream: int) -> None:
        _lib.check(self._lib.ludwig_level_set_stream(self.handle, C.c_void_p(hip_stream)))

    def set_order(self, items: np.ndarray, part: int = _lib.PART_ALL) -> None:
        items = np.ascontiguousarray(items, dtype=np.int32)
        _lib.check(self._lib.ludwig_level_set_order(self.handle, part, items.ctypes.data, items.size))

    # -- field access --
    def _shape(self, name: str):
        B, n = BLOCK_SIZE, self.n_blocks
        if name in ("f", "f_temp", "f_post_collision", "f_old"):
            return (B, B, B, n, 27)
        if name in ("vel", "vel_temp", "vel_old"):
            return (B, B, B, n, 3)
        return (B, B, B, n)

    def upload(self, name: str, host: np.ndarray) -> None:
        dt = _FIELD_DTYPES.get(name, np.float32)
        a = host.view(np.uint8) if host.dtype == np.bool_ else host
        a = np.asfortranarray(a, dtype=dt)
        if a.shape != self._shape(name):
            raise ValueError(f"{name}: shape {a.shape} != {self._shape(name)}")
        _lib.check(self._lib.ludwig_level_upload(self.handle, _lib.FIELD_NAMES[name], a.ctypes.data, a.nbytes))

    def download(self, name: str) -> np.ndarray:
        dt = _FIELD_DTYPES.get(name, np.float32)
        a = np.empty(self._shape(name), dtype=dt, order="F")
        _lib.check(self._lib.ludwig_level_download(self.handle, _lib.FIELD_NAMES[name], a.ctypes.data, a.nbytes))
        return a.view(np.bool_) if name == "obstacle" else a

    def block_order(self) -> np.ndarray:
        """ref_to_internal [n_blocks]: where the library keeps block b of the reference order (only raw pointers show it)"""
        a = np.zeros(self.n_blocks, dtype=np.int32)
        _lib.check(self._lib.ludwig_level_block_order(self.handle, a.ctypes.data))
        return a

    def field_ptr(self, name: str) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t()
        _lib.check(self._lib.ludwig_level_field_ptr(self.handle, _lib.FIELD_NAMES[name], C.byref(p), C.byref(n)))
        return p.value, n.value

    def field_layout(self, name: str) -> Tuple[int, int, int]:
        """(components K, block stride, component stride) of a device array in elements - block-major storage; only raw pointers see it"""
        k, bs, cs = C.c_int32(), C.c_int64(), C.c_int64()
        _lib.check(self._lib.ludwig_level_field_layout(self.handle, _lib.FIELD_NAMES[name], C.byref(k), C.byref(bs), C.byref(cs)))
        return int(k.value), int(bs.value), int(cs.value)

    def set_rho_store(self, every_step: bool) -> None:
        """True: rho is stored by every step like the reference's kernel (src/physics_kernels.jl:243-246); False: elided when unread"""
        _lib.check(self._lib.ludwig_level_set_rho_store(self.handle, 1 if every_step else 0))

    def rho_min(self) -> float:
        """rho_min of compute_flow_stats (src/diagnostics.jl:56-94) over the owned, non-obstacle cells, reduced on the device"""
        v = C.c_float()
        _lib.check(self._lib.ludwig_level_rho_min(self.handle, C.byref(v)))
        return float(v.value)

    def init_equilibrium(self) -> None:
        """init_eq! (src/main.jl:109-134)"""
        _lib.check(self._lib.ludwig_init_equilibrium(self.handle))

    def copy_to_old(self, t_sub: int) -> None:
        """copy_to_old!(level, f_in, vel_in) for the step t_sub about to run (src/blocks.jl:199-205)."""
        _lib.check(self._lib.ludwig_save_old(self.handle, int(t_sub)))

    def synchronize(self) -> None:
        _lib.check(self._lib.ludwig_sync(self.handle))


def adapt(host: BlockLevel, device: int = 0, upload_state: bool = True) -> DeviceLevel:
    """grids = [adapt(backend, g) for g in cpu_grids] (src/main.jl:98)"""
    return DeviceLevel(host, device, upload_state)
