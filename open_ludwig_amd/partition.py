"""Static spatial partition of one BlockLevel across GPUs + the one-cell halo exchange.

The reference is single-device (F2); this module has no reference counterpart. Design:

  * every rank holds a LOCAL BlockLevel = its owned blocks (in global order) followed by GHOST copies of the remote
    blocks any owned block touches. `neighbor_table` stays the only addressing mechanism, so the kernels do not change.
  * after a step the ghost cells the next step will read are refreshed: the inward populations of `f_out` on the
    one-cell face layer and the three components of `vel_out` (WALE stencil). What exactly is read is DERIVED from
    the pull / stencil rules, cell by cell, into index lists ("needs"); the owner of each block serves them.
  * exchange = gather by index list -> one message per peer (RCCL send/recv, point-to-point over xGMI) -> scatter.
    No collective on the data path. The exchange runs on a second HIP stream under compute: a single level (and the finest
    of nested levels) steps its interior blocks - which read no ghost - first, under the exchange of the PREVIOUS step, then
    its boundary blocks; a level with children steps its boundary blocks first and exchanges under its interior blocks,
    because its children need the ghosts next. Levels with Bouzidi cells fit in: their small f_post halo is waited for
    between the collision and the correction. The RCCL backend has never executed any of this (one-GPU boxes): rehearsed
    over gloo with host staging, all ranks on one device.

  * nested levels (row N3): every level is cut on its own into equal parts (level_owners); a rank's copy of a level also
    holds ghost copies of the parent blocks its finer blocks interpolate from (required_parent_blocks), and the parent
    cells of those interface stencils join the parent level's needs (interpolation_needs). MultiLevelRunner repeats the
    reference's recursion with one exchange after every level step.
  * diagnostics cross ranks as scalars only (case.DistributedStepper: rho_min by all-reduce MIN, nine partial force sums
    per rank and one all-gather); whole fields are gathered on result-file steps, to rank 0.

Pure-numpy planning (build_local_level, compute_needs, interpolation_needs, HaloPlan) is separate from transport
(HaloExchanger) and from the GPU runners, so the N > 1 logic is covered by CPU tests over gloo.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .blocks import BLOCK_SIZE, BlockLevel, build_lattice_arrays, build_neighbor_table

_CX, _CY, _CZ, _W, _OPP, _MY, _MZ = build_lattice_arrays()
CELLS = 512
FIELD_GROUPS = ("f", "vel", "f_post", "rho")   # logical halo fields: populations, velocity, post-collision populations, density


# ----------------------------------------------------------------------------------------------------------------
# planning (numpy only)
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class LocalView:
    """One rank's slice of a global level."""
    rank: int
    level: BlockLevel                 # local level: owned blocks then ghosts; map_x/y/z keep GLOBAL block coords
    local_to_global: np.ndarray       # [n_local] global block id (0-based)
    global_to_local: Dict[int, int]   # only for blocks present locally
    n_owned: int
    ghost_owner: np.ndarray           # [n_local - n_owned] owning rank of every ghost block


def build_local_level(level_id: int, coords: Sequence[Tuple[int, int, int]], neighbor_table: np.ndarray, owner: np.ndarray,
                      rank: int, tau: float, temporal: bool = False, extra_ghosts: Optional[np.ndarray] = None,
                      widen_x_runs: bool = False) -> LocalView:
    """Cut rank `rank`'s local level out of the global block list (coords + global neighbor_table, 1-based).
    extra_ghosts: further remote blocks (global ids, 0-based) to keep a ghost copy of - parent data of this rank's finer
    blocks (required_parent_blocks)."""
    coords = np.asarray(coords, dtype=np.int64).reshape(-1, 3)
    owner = np.asarray(owner)
    nt = np.asarray(neighbor_table)
    owned = np.flatnonzero(owner == rank)
    nbrs = nt[owned].reshape(-1)
    nbrs = np.unique(nbrs[nbrs > 0]) - 1
    ghosts = nbrs[owner[nbrs] != rank]
    if extra_ghosts is not None and len(extra_ghosts):
        extra = np.asarray(extra_ghosts, dtype=np.int64)
        ghosts = np.unique(np.concatenate([ghosts, extra[owner[extra] != rank]]))
    l2g = np.concatenate([owned, ghosts])
    g2l = np.full(len(coords) + 1, 0, dtype=np.int64)            # 1-based global -> 1-based local, 0 = absent
    g2l[l2g + 1] = np.arange(1, len(l2g) + 1)
    table = np.zeros((len(l2g), 27), dtype=np.int32, order="F")
    table[: len(owned)] = g2l[nt[owned]]                            # ghost rows stay 0: ghosts are never stepped
    lvl = BlockLevel(level_id, [tuple(c) for c in coords[l2g]], table, 1.0, 1.0, tau, enable_temporal_interpolation=temporal)
    lvl.n_owned = len(owned)
    cb = np.zeros(len(l2g), dtype=np.uint8)
    cb[: len(owned)] = (table[: len(owned)] > len(owned)).any(axis=1)
    if widen_x_runs and len(owned):
        # A boundary block on an x face has its x neighbours in the interior part: alone in its launch it cannot join an x-run
        # (the workgroup of 4 x-consecutive blocks that hands face columns over through LDS) and reads them as strided columns.
        # Taking the whole aligned group of 4 into the boundary part keeps the runs whole; a superset of the blocks that must
        # wait for the halo is always valid.
        oc = coords[owned]
        key = ((oc[:, 0] - 1) // 4) * (int(coords[:, 1].max()) + 1) * (int(coords[:, 2].max()) + 1) + oc[:, 1] * (int(coords[:, 2].max()) + 1) + oc[:, 2]
        marked = np.unique(key[cb[: len(owned)] != 0])
        cb[: len(owned)] = np.isin(key, marked)
    lvl.comm_boundary = cb
    return LocalView(rank, lvl, l2g, {int(g): i for i, g in enumerate(l2g)}, len(owned), owner[ghosts])


def _take_blocks(a: np.ndarray, g: np.ndarray) -> np.ndarray:
    """a[:, :, :, g] (and a[:, :, :, g, :]) of a Fortran-ordered level array as whole 512-cell chunks: the transposed view is
    C-contiguous [K][block][z][y][x], so the gather moves 2-KiB pieces instead of single elements"""
    if not a.flags.f_contiguous:
        return np.asfortranarray(a[:, :, :, g])
    n = a.shape[3]
    if a.ndim == 4:
        return np.take(a.T.reshape(n, 512), g, axis=0).reshape(len(g), 8, 8, 8).T
    k = a.shape[4]
    return np.take(a.T.reshape(k, n, 512), g, axis=1).reshape(k, len(g), 8, 8, 8).T


def slice_level_fields(view: LocalView, source: BlockLevel, state: bool = True) -> None:
    """Fill a local level (owned + ghost blocks) from a populated GLOBAL host level: state, geometry, sponge, wall
    distance and the Bouzidi data of the owned blocks (cell list re-indexed to local block ids).
    state=False: geometry only - for callers that initialise the state on the device afterwards (DistributedStepper: init_eq!)."""
    lvl, g = view.level, view.local_to_global
    def take(name):
        a = _take_blocks(getattr(source, name), g)          # a new Fortran-ordered array of the local level's shape: it replaces the
        assert a.shape == getattr(lvl, name).shape and a.dtype == getattr(lvl, name).dtype, name       # default one (no second copy)
        setattr(lvl, name, a)

    for name in (("rho", "vel", "vel_temp", "f", "f_temp") if state else ()) + ("obstacle", "sponge", "wall_dist"):
        take(name)
    if state and lvl.f_old.size > 27 and source.f_old.size > 27:
        for name in ("f_old", "rho_old", "vel_old"):
            take(name)
    if source.bouzidi_enabled:
        g2l = np.full(source.n_blocks, -1, dtype=np.int64)
        g2l[g[: view.n_owned]] = np.arange(view.n_owned)
        lb = g2l[source.bouzidi_cell_block.astype(np.int64) - 1]
        keep = lb >= 0
        B = BLOCK_SIZE
        lvl.bouzidi_q_map = _take_blocks(source.bouzidi_q_map, g)
        lvl.bouzidi_cell_block = (lb[keep] + 1).astype(np.int32)
        lvl.bouzidi_cell_x = source.bouzidi_cell_x[keep].copy()
        lvl.bouzidi_cell_y = source.bouzidi_cell_y[keep].copy()
        lvl.bouzidi_cell_z = source.bouzidi_cell_z[keep].copy()
        lvl.n_boundary_cells = int(keep.sum())
        lvl.bouzidi_enabled = lvl.n_boundary_cells > 0
        # every rank of a Bouzidi level stores f_post_collision (its ghosts may be asked for it)
        if state and source.f_post_collision.size > 27:
            lvl.f_post_collision = _take_blocks(source.f_post_collision, g)
        else:
            lvl.f_post_collision = np.zeros((B, B, B, lvl.n_blocks, 27), dtype=np.float32, order="F")
        # keep the store of f_post_collision alive in every block of this rank: a peer's Bouzidi cells may read our face layer
        lvl.force_post_collision = True


def _source_of(cx: int, cy: int, cz: int):
    """for every cell of an 8^3 block: (dir index 0..26 of the block holding cell - c, cell index inside that block)"""
    B = BLOCK_SIZE
    x, y, z = np.meshgrid(np.arange(B), np.arange(B), np.arange(B), indexing="ij")
    sx, sy, sz = x - cx, y - cy, z - cz
    ox = np.where(sx < 0, -1, np.where(sx >= B, 1, 0))
    oy = np.where(sy < 0, -1, np.where(sy >= B, 1, 0))
    oz = np.where(sz < 0, -1, np.where(sz >= B, 1, 0))
    d = (ox + 1) + 3 * (oy + 1) + 9 * (oz + 1)
    cell = (sx % B) + B * (sy % B) + B * B * (sz % B)
    own = x + B * y + B * B * z
    o = np.argsort(own.reshape(-1))
    return d.reshape(-1)[o], cell.reshape(-1)[o]


def compute_needs(view: LocalView) -> Dict[str, np.ndarray]:
    """Ghost elements the next step of the OWNED blocks will read, as local element offsets (reference layout):
      'f'   : offsets into f   [8,8,8,nb,27]  - pull rule of src/physics_kernels.jl:62-86
      'vel' : offsets into vel [8,8,8,nb,3]   - face stencil of src/physics_utils.jl:45-83
    plus for each the ghost block (local id) every element lives in."""
    lvl = view.level
    nb, n_owned = lvl.n_blocks, view.n_owned
    table = np.asarray(lvl.neighbor_table)[:n_owned]                    # 1-based local ids
    bnd = np.flatnonzero(lvl.comm_boundary[:n_owned])
    out_f: List[np.ndarray] = []
    out_v: List[np.ndarray] = []
    if bnd.size:
        tb = table[bnd]                                                 # [n_bnd, 27]
        for k in range(27):
            d, cell = _source_of(int(_CX[k]), int(_CY[k]), int(_CZ[k]))
            cross = np.flatnonzero(d != 13)
            if cross.size == 0:
                continue
            src_blk = tb[:, d[cross]]                                   # [n_bnd, n_cross] 1-based local ids
            m = src_blk > n_owned
            if m.any():
                blk0 = src_blk[m].astype(np.int64) - 1
                cells = np.broadcast_to(cell[cross], src_blk.shape)[m]
                out_f.append((k * nb + blk0) * CELLS + cells)
        for (dx, dy, dz) in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
            d, cell = _source_of(-dx, -dy, -dz)                          # neighbour at +d == source of c = -d
            cross = np.flatnonzero(d != 13)
            src_blk = tb[:, d[cross]]
            m = src_blk > n_owned
            if m.any():
                blk0 = src_blk[m].astype(np.int64) - 1
                cells = np.broadcast_to(cell[cross], src_blk.shape)[m]
                for comp in range(3):
                    out_v.append((comp * nb + blk0) * CELLS + cells)
    f = np.unique(np.concatenate(out_f)) if out_f else np.zeros(0, np.int64)
    v = np.unique(np.concatenate(out_v)) if out_v else np.zeros(0, np.int64)
    needs = {"f": f, "vel": v}
    if lvl.bouzidi_enabled and lvl.n_boundary_cells > 0:
        # Bouzidi q < 1/2 reads f_post_collision[k] at cell + c_opp(k) (src/bouzidi_kernel.jl:44-77); across a cut that
        # cell lives in a ghost block. All 0 < q < 1/2 links are listed (a superset of q > q_min: harmless).
        B = BLOCK_SIZE
        cb = lvl.bouzidi_cell_block.astype(np.int64) - 1
        x = lvl.bouzidi_cell_x.astype(np.int64) - 1; y = lvl.bouzidi_cell_y.astype(np.int64) - 1; z = lvl.bouzidi_cell_z.astype(np.int64) - 1
        out_p: List[np.ndarray] = []
        full_table = np.asarray(lvl.neighbor_table)
        for k in range(27):
            q = lvl.bouzidi_q_map[x, y, z, cb, k].astype(np.float32)
            m = (q > 0) & (q < 0.5)
            if not m.any():
                continue
            ok = int(_OPP[k]) - 1
            nx, ny, nz = x[m] + int(_CX[ok]), y[m] + int(_CY[ok]), z[m] + int(_CZ[ok])
            ox = np.where(nx < 0, -1, np.where(nx >= B, 1, 0)); oy = np.where(ny < 0, -1, np.where(ny >= B, 1, 0)); oz = np.where(nz < 0, -1, np.where(nz >= B, 1, 0))
            d = (ox + 1) + 3 * (oy + 1) + 9 * (oz + 1)
            nbl = full_table[cb[m], d].astype(np.int64)          # 1-based local id
            gh = nbl > n_owned
            if gh.any():
                cell = (nx[gh] % B) + B * (ny[gh] % B) + B * B * (nz[gh] % B)
                out_p.append((k * nb + nbl[gh] - 1) * CELLS + cell)
        needs["f_post"] = np.unique(np.concatenate(out_p)) if out_p else np.zeros(0, np.int64)
    return needs


def _to_global(view: LocalView, local_off: np.ndarray, n_global: int) -> np.ndarray:
    nb = view.level.n_blocks
    comp, rem = np.divmod(local_off, nb * CELLS)
    blk, cell = np.divmod(rem, CELLS)
    return (comp * n_global + view.local_to_global[blk]) * CELLS + cell


def _to_local(view: LocalView, global_off: np.ndarray, n_global: int) -> np.ndarray:
    nb = view.level.n_blocks
    comp, rem = np.divmod(global_off, n_global * CELLS)
    gblk, cell = np.divmod(rem, CELLS)
    g2l = np.full(n_global, -1, dtype=np.int64)
    g2l[view.local_to_global] = np.arange(nb)
    lblk = g2l[gblk]
    assert (lblk >= 0).all() and (lblk < view.n_owned).all(), "peer asked for a block this rank does not own"
    return (comp * nb + lblk) * CELLS + cell


@dataclass
class HaloPlan:
    """Per peer: what to receive (local offsets into ghost blocks) and what to send (local offsets into owned blocks),
    per logical field (FIELD_GROUPS). Receive lists are sorted by global offset, and send lists follow the receiver's order."""
    peers: List[int] = field(default_factory=list)
    recv: Dict[int, Dict[str, np.ndarray]] = field(default_factory=dict)
    send: Dict[int, Dict[str, np.ndarray]] = field(default_factory=dict)

    def bytes_per_step(self) -> int:
        return 4 * sum(a.size for p in self.peers for a in self.send[p].values())

    def has(self, name: str) -> bool:
        return any(self.send[p][name].size or self.recv[p][name].size for p in self.peers)


def make_requests(view: LocalView, n_global: int, needs: Optional[Dict[str, np.ndarray]] = None) -> Dict[int, Dict[str, np.ndarray]]:
    """What this rank asks of every peer: global element offsets, sorted."""
    if needs is None:
        needs = compute_needs(view)
    nb, n_owned = view.level.n_blocks, view.n_owned
    req: Dict[int, Dict[str, np.ndarray]] = {}
    for name, off in needs.items():
        blk = (off % (nb * CELLS)) // CELLS
        own = view.ghost_owner[blk - n_owned]
        for p in np.unique(own):
            sel = off[own == p]
            g = _to_global(view, sel, n_global)
            o = np.argsort(g, kind="stable")
            req.setdefault(int(p), {})[name] = g[o]
    for p in req:
        for name in FIELD_GROUPS:
            req[p].setdefault(name, np.zeros(0, np.int64))
    return req


def build_plan(view: LocalView, n_global: int, my_requests: Dict[int, Dict[str, np.ndarray]],
               requests_to_me: Dict[int, Dict[str, np.ndarray]]) -> HaloPlan:
    plan = HaloPlan()
    plan.peers = sorted(set(my_requests) | set(requests_to_me))
    nb = view.level.n_blocks
    g2l = np.full(n_global, -1, dtype=np.int64)
    g2l[view.local_to_global] = np.arange(nb)
    for p in plan.peers:
        plan.recv[p] = {}
        plan.send[p] = {}
        for name in FIELD_GROUPS:
            g = my_requests.get(p, {}).get(name, np.zeros(0, np.int64))
            comp, rem = np.divmod(g, n_global * CELLS)
            gblk, cell = np.divmod(rem, CELLS)
            plan.recv[p][name] = (comp * nb + g2l[gblk]) * CELLS + cell
            plan.send[p][name] = _to_local(view, requests_to_me.get(p, {}).get(name, np.zeros(0, np.int64)), n_global)
    return plan


# ----------------------------------------------------------------------------------------------------------------
# transport (torch.distributed): pack -> send/recv -> unpack with pluggable pack/unpack
# ----------------------------------------------------------------------------------------------------------------
class HaloExchanger:
    """One exchange = pack(all send lists) -> per peer isend/irecv of slices -> unpack(all recv lists).

    pack(name, index_tensor, out_tensor) / unpack(name, index_tensor, in_tensor) are supplied by the caller:
    HIP gather/scatter kernels on the GPU path, numpy fancy indexing in the CPU (gloo) tests. All peers' lists of one
    field are concatenated, so a step costs 2 pack + 2 unpack launches and one grouped send/recv, however many peers.
    `device` is where the message buffers live; with `stage_through_host` (backend gloo, GPU buffers) the messages go
    through pinned host memory - used to rehearse > 1 rank on a single GPU, where RCCL refuses to start.
    """

    def __init__(self, plan: HaloPlan, rank: int, device, pack: Callable, unpack: Callable, stage_through_host: bool = False):
        import torch
        self.torch = torch
        self.plan, self.rank = plan, rank
        self.pack, self.unpack = pack, unpack
        self.stage = stage_through_host
        self.device = torch.device(device)
        self.timing = False                # bench.py: bracket every exchange with events on the stream it runs on
        # peer of the plan -> rank the message really travels to / from. Identity in production. The RCCL loop-back test
        # (tests/_rccl_loopback_worker.py) plays rank 0 of a brick decomposition whose bricks hold identical data and maps every
        # peer to rank 0 itself, so that the real transport runs on a box with one GPU.
        self.wire_rank: Dict[int, int] = {}
        self._events = []
        self.idx_send, self.idx_recv, self.buf_send, self.buf_recv, self.seg_send, self.seg_recv = {}, {}, {}, {}, {}, {}
        for name in FIELD_GROUPS:
            s_lists = [plan.send[p][name] for p in plan.peers]
            r_lists = [plan.recv[p][name] for p in plan.peers]
            cat = lambda ls: np.concatenate(ls) if ls else np.zeros(0, np.int64)
            self.idx_send[name] = torch.as_tensor(cat(s_lists), dtype=torch.int64, device=device)
            self.idx_recv[name] = torch.as_tensor(cat(r_lists), dtype=torch.int64, device=device)
            self.buf_send[name] = torch.empty(self.idx_send[name].numel(), dtype=torch.float32, device=device)
            self.buf_recv[name] = torch.empty(self.idx_recv[name].numel(), dtype=torch.float32, device=device)
            so = np.concatenate([[0], np.cumsum([len(a) for a in s_lists])]).astype(np.int64)
            ro = np.concatenate([[0], np.cumsum([len(a) for a in r_lists])]).astype(np.int64)
            self.seg_send[name] = {p: (int(so[i]), int(so[i + 1])) for i, p in enumerate(plan.peers)}
            self.seg_recv[name] = {p: (int(ro[i]), int(ro[i + 1])) for i, p in enumerate(plan.peers)}
        if self.stage:
            self.host_send = {n: torch.empty(self.buf_send[n].numel(), dtype=torch.float32).pin_memory() for n in FIELD_GROUPS}
            self.host_recv = {n: torch.empty(self.buf_recv[n].numel(), dtype=torch.float32).pin_memory() for n in FIELD_GROUPS}

    def exchange(self, f_name: str, vel_name: str) -> None:
        """Refresh the ghost elements of fields `f_name` ('f' | 'f_temp') and `vel_name` ('vel' | 'vel_temp')."""
        self.exchange_fields({"f": f_name, "vel": vel_name})

    def exchange_post_collision(self) -> None:
        """Refresh the ghost elements of f_post_collision that Bouzidi cells next to a partition cut read."""
        self.exchange_fields({"f_post": "f_post_collision"})

    def exchange_fields(self, fields: Dict[str, str]) -> None:
        """fields: logical halo group ('f' | 'vel' | 'f_post' | 'rho') -> name of the level field to move."""
        import torch.distributed as dist
        torch = self.torch
        names = tuple(fields)
        on_gpu = self.device.type == "cuda"
        if self.timing and on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        for n in names:
            if self.buf_send[n].numel():
                self.pack(fields[n], self.idx_send[n], self.buf_send[n])
        if self.stage:
            for n in names:
                self.host_send[n].copy_(self.buf_send[n], non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        snd = self.host_send if self.stage else self.buf_send
        rcv = self.host_recv if self.stage else self.buf_recv
        ops = []
        for p in self.plan.peers:
            for n in names:
                a, b = self.seg_send[n][p]
                c, d = self.seg_recv[n][p]
                if p == self.rank:
                    if d > c:
                        rcv[n][c:d].copy_(snd[n][a:b])
                    continue
                w = self.wire_rank.get(p, p)
                if b > a:
                    ops.append(dist.P2POp(dist.isend, snd[n][a:b], w))
                if d > c:
                    ops.append(dist.P2POp(dist.irecv, rcv[n][c:d], w))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.stage:
            for n in names:
                self.buf_recv[n].copy_(self.host_recv[n], non_blocking=True)
        for n in names:
            if self.buf_recv[n].numel():
                self.unpack(fields[n], self.idx_recv[n], self.buf_recv[n])
        if self.timing and on_gpu:
            e1.record(torch.cuda.current_stream(self.device))
            self._events.append((e0, e1))

    def exchange_ms(self) -> List[float]:
        """elapsed time of every timed exchange (pack -> send/recv -> unpack) on its stream; call after a device synchronize"""
        out = [a.elapsed_time(b) for a, b in self._events]
        self._events = []
        return out


class NativeHalo:
    """The halo exchange of one level behind the C ABI (include/ludwig_hip.h "multi-GPU"): the plan is translated and uploaded once
    (ludwig_halo_plan_create), an exchange is one call - pack, one grouped ncclSend / ncclRecv per peer straight from RCCL, unpack, all
    on the plan's own high-priority stream - and costs the host three kernel launches and a group call. `comm`: a handle from
    native_comm(), or None when every peer is this rank itself (device copies). wire_rank: peer of the plan -> rank the message
    really travels to (the loop-back test wires every peer to rank 0)."""

    def __init__(self, plan: HaloPlan, level, comm, wire_rank: Optional[Dict[int, int]] = None):
        import ctypes as C
        from . import _lib
        self._lib, self.C = _lib, C
        self.plan, self.level, self.comm = plan, level, comm
        lib = _lib.load()
        n = len(plan.peers)
        wire = wire_rank or {}
        peers = np.asarray([wire.get(p, p) for p in plan.peers], dtype=np.int32)
        d = _lib.HaloPlanDesc()
        d.n_peers = n
        d.peer_ranks = peers.ctypes.data
        keep = [peers]
        for gi, name in enumerate(FIELD_GROUPS):
            sc = np.asarray([len(plan.send[p][name]) for p in plan.peers], dtype=np.int64)
            rc = np.asarray([len(plan.recv[p][name]) for p in plan.peers], dtype=np.int64)
            si = np.ascontiguousarray(np.concatenate([plan.send[p][name] for p in plan.peers]) if n else np.zeros(0), dtype=np.int64)
            ri = np.ascontiguousarray(np.concatenate([plan.recv[p][name] for p in plan.peers]) if n else np.zeros(0), dtype=np.int64)
            keep += [sc, rc, si, ri]
            d.send_count[gi], d.recv_count[gi] = sc.ctypes.data, rc.ctypes.data
            d.send_index[gi], d.recv_index[gi] = si.ctypes.data, ri.ctypes.data
        h = C.c_void_p()
        _lib.check(lib.ludwig_halo_plan_create(level.handle, comm, C.byref(d), C.byref(h)))
        self._h = h
        self._timing = False
        self.in_stream = False

    @property
    def handle(self):
        return self._h

    def post(self, fields: Dict[str, str]) -> None:
        """queue one exchange behind everything queued on the level's stream so far; returns at once"""
        C, _lib = self.C, self._lib
        n = len(fields)
        groups = (C.c_int32 * n)(*[FIELD_GROUPS.index(g) for g in fields])
        flds = (C.c_int32 * n)(*[_lib.FIELD_NAMES[f] for f in fields.values()])
        _lib.check(_lib.load().ludwig_halo_exchange(self._h, n, groups, flds))

    def join(self) -> None:
        """what is queued on the level's stream from here on runs after the last posted exchange"""
        self._lib.check(self._lib.load().ludwig_halo_wait(self._h))

    def set_in_stream(self, on: bool) -> None:
        """the exchange on the level's own stream (no overlap, no cross-stream hand-over): ludwig_halo_plan_in_stream"""
        self._lib.check(self._lib.load().ludwig_halo_plan_in_stream(self._h, 1 if on else 0))
        self.in_stream = bool(on)

    @property
    def timing(self) -> bool:
        return self._timing

    @timing.setter
    def timing(self, on: bool) -> None:
        self._lib.check(self._lib.load().ludwig_halo_plan_timing(self._h, 1 if on else 0))
        self._timing = bool(on)

    def exchange_ms(self) -> List[float]:
        C = self.C
        out: List[float] = []
        buf = (C.c_float * 256)()
        n = C.c_int32(0)
        while True:
            self._lib.check(self._lib.load().ludwig_halo_plan_exchange_ms(self._h, buf, 256, C.byref(n)))
            out += [float(buf[i]) for i in range(n.value)]
            if n.value < 256:
                return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.load().ludwig_halo_plan_destroy(self._h)
            self._h = None


class TorchHalo:
    """The same post / join interface over torch.distributed (HaloExchanger): the transport of the one-GPU rehearsals (gloo with
    host-staged messages; RCCL refuses two ranks on one device) and of the CPU tests."""

    def __init__(self, ex: "HaloExchanger", s_comp, s_comm):
        import torch
        self.torch, self.ex, self.plan = torch, ex, ex.plan
        self.s_comp, self.s_comm = s_comp, s_comm
        self.ev_ready, self.ev_done = torch.cuda.Event(), torch.cuda.Event()
        self.pending = False

    def post(self, fields: Dict[str, str]) -> None:
        if self.s_comm is self.s_comp:
            self.ex.exchange_fields(fields)
            return
        self.ev_ready.record(self.s_comp)
        with self.torch.cuda.stream(self.s_comm):
            self.s_comm.wait_event(self.ev_ready)
            self.ex.exchange_fields(fields)
            self.ev_done.record(self.s_comm)
        self.pending = True

    def join(self) -> None:
        if self.pending:
            self.s_comp.wait_event(self.ev_done)
            self.pending = False

    @property
    def timing(self) -> bool:
        return self.ex.timing

    @timing.setter
    def timing(self, on: bool) -> None:
        self.ex.timing = on

    def exchange_ms(self) -> List[float]:
        return self.ex.exchange_ms()

    def close(self) -> None:
        self.ex._events = []
        self.ev_ready = self.ev_done = None


def native_comm(device: int):
    """A LudwigComm over the ranks of the default torch.distributed group: rank 0 draws the RCCL unique id (ludwig_comm_unique_id),
    the 128 bytes travel through the process group the ranks already share, every rank joins (ludwig_comm_create). Returns the
    handle (a ctypes.c_void_p); None when torch.distributed is not initialised (single process: every peer is this rank).
    Raises LudwigError on EVERY rank when any rank could not come up (the ranks agree on the outcome before anybody proceeds)."""
    import ctypes as C
    import torch.distributed as dist
    from . import _lib
    if not dist.is_initialized():
        return None
    lib = _lib.load()
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [None]
    if rank == 0:
        buf = (C.c_char * _lib.UNIQUE_ID_BYTES)()
        rc = lib.ludwig_comm_unique_id(buf)
        box[0] = bytes(buf) if rc == 0 else ("error", (lib.ludwig_last_error() or b"").decode("utf-8", "replace"))
    dist.broadcast_object_list(box, src=0)
    if isinstance(box[0], tuple):
        raise _lib.LudwigError(-5, "rank 0 could not draw an RCCL unique id: " + box[0][1])
    h = C.c_void_p()
    rc = lib.ludwig_comm_create(box[0], rank, world, int(device), C.byref(h))
    msg = (lib.ludwig_last_error() or b"").decode("utf-8", "replace") if rc else ""
    oks = [None] * world
    dist.all_gather_object(oks, (rc, msg))
    bad = [(r, m) for r, (c, m) in enumerate(oks) if c != 0]
    if bad:
        if rc == 0:
            lib.ludwig_comm_destroy(h)
        raise _lib.LudwigError(bad[0][1] and -2 or -2, f"ludwig_comm_create failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}")
    return h


def native_comm_destroy(comm) -> None:
    if comm:
        from . import _lib
        _lib.load().ludwig_comm_destroy(comm)


def init_rccl(local_rank: int) -> None:
    """`torch.distributed` over RCCL with RCCL's kernels on a HIGH-priority stream. Not a tuning nicety: HIP serves the streams of one
    priority from a small pool of hardware queues, and a send/recv kernel that lands in the queue of the compute stream starts only
    when the stream-collide launch ahead of it has drained - the exchange is then not overlapped at all (seen in the loop-back
    trace: profiles/r02_rccl_loopback_trace_same_queue.txt). Streams of another priority get queues of their own."""
    import torch
    import torch.distributed as dist
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)


def exchange_requests(my_requests: Dict[int, Dict[str, np.ndarray]], world: int, rank: int) -> Dict[int, Dict[str, np.ndarray]]:
    """Setup-time hand-shake: every rank learns what the others want from it (object all-gather; works on gloo and nccl)."""
    import torch.distributed as dist
    gathered: List[Optional[dict]] = [None] * world
    dist.all_gather_object(gathered, my_requests)
    to_me: Dict[int, Dict[str, np.ndarray]] = {}
    for r, reqs in enumerate(gathered):
        if reqs and rank in reqs:
            to_me[r] = reqs[rank]
    return to_me


# ----------------------------------------------------------------------------------------------------------------
# global topologies used by bench.py and the tests
# ----------------------------------------------------------------------------------------------------------------
def rank_grid(world: int) -> Tuple[int, int, int]:
    """1 -> (1,1,1), 2 -> (1,1,2), 4 -> (1,2,2), 8 -> (2,2,2): bricks, so faces are 1/4 the size of slabs' (SURVEY 8e).
    z is cut first and x last: a z face is whole 256-B rows of every population (the pack / unpack kernels move full sectors) and
    leaves the runs of x-consecutive blocks the stepping kernel works in whole; an x face is single cells 32 B apart and lone blocks
    (RCCL loop-back, 256^3 per brick: 1x1x2 0.759 vs 2x1x1 0.79-0.83 ms per step, 1x2x2 0.786 vs 2x2x1 0.81-0.83)."""
    g = [1, 1, 1]
    a = 0
    w = world
    while w > 1:
        if w % 2:
            raise ValueError("world size must be a power of two")
        g[2 - a % 3] *= 2
        w //= 2
        a += 1
    return tuple(g)


def periodic_box_topology(nb_global: Tuple[int, int, int], grid: Tuple[int, int, int]):
    """coords (reference order), periodic neighbor_table and the brick owner of every block."""
    nbx, nby, nbz = nb_global
    coords = [(bx, by, bz) for bx in range(1, nbx + 1) for by in range(1, nby + 1) for bz in range(1, nbz + 1)]
    table = build_neighbor_table(coords, nbx, nby, nbz, (True, True, True))
    c = np.asarray(coords) - 1
    per = (nbx // grid[0], nby // grid[1], nbz // grid[2])
    owner = ((c[:, 0] // per[0]) * grid[1] + (c[:, 1] // per[1])) * grid[2] + (c[:, 2] // per[2])
    return coords, table, owner.astype(np.int64)


def name_post_collision_readers(level: BlockLevel, plan: "HaloPlan") -> None:
    """A rank of a Bouzidi level stores f_post_collision for its own links and for the links of its PEERS that reach across a cut -
    and those are exactly this rank's f_post send lists. Hands them to the level (`post_collision_readers`, read by DeviceLevel) so
    that the step stores the rows with a reader instead of every block (`force_post_collision`: 108 B per cell update more)."""
    if not getattr(level, "force_post_collision", False) or os.environ.get("LUDWIG_FULL_POST_COLLISION"):
        return
    offs = [np.asarray(plan.send[pr]["f_post"], dtype=np.int64) for pr in plan.peers if plan.has("f_post") and len(plan.send[pr].get("f_post", ()))]
    level.post_collision_readers = np.concatenate(offs) if offs else np.zeros(0, np.int64)


class DistributedLevelRunner:
    """GPU path: one rank's local level on one MI355X + halo exchange; step(t) = one stream-collide pass everywhere.

    transport "native" (default): the exchange and the overlap schedule live behind the C ABI - ludwig_step_distributed: interior
    blocks, wait for the previous exchange, boundary blocks, [f_post halo, Bouzidi correction], this step's exchange left in flight -
    with RCCL called from the library (NativeHalo). transport "torch" (always with stage_through_host): the same schedule driven from
    here over torch.distributed (TorchHalo) - the one-GPU rehearsals over gloo.
    overlap=False: whole-level launch, then the exchange, waited for."""

    def __init__(self, view: LocalView, plan: HaloPlan, params, device: int, overlap: bool = True,
                 stage_through_host: bool = False, order: Optional[str] = None, transport: Optional[str] = None, comm=None,
                 wire_rank: Optional[Dict[int, int]] = None):
        import ctypes as C
        import torch
        from . import _lib
        from .blocks import adapt
        from . import order as order_mod
        self.torch, self._lib, self.C = torch, _lib, C
        self.view, self.params, self.overlap = view, params, overlap
        self.transport = transport or ("torch" if stage_through_host else os.environ.get("LUDWIG_HALO_TRANSPORT", "native"))
        assert self.transport in ("native", "torch") and not (stage_through_host and self.transport == "native")
        torch.cuda.set_device(device)
        name_post_collision_readers(view.level, plan)
        self.level = adapt(view.level, device)
        self.dev = torch.device("cuda", device)
        # overlap: the stepping stream leaves compute units to the exchange (include/ludwig_hip.h: ludwig_stream_create;
        # LUDWIG_COMM_RESERVED_CUS, 0 = none); pack / unpack and RCCL run on a high-priority stream. 32 = one CU out of every shader
        # engine of every XCD: the cheapest mask there is (the library rounds any request up to that, profiles/r03_cu_mask_patterns.txt)
        self.reserved_cus = int(os.environ.get("LUDWIG_COMM_RESERVED_CUS", "32")) if overlap else 0
        if self.reserved_cus > 0:
            self.reserved_cus = -(-self.reserved_cus // 32) * 32          # what ludwig_stream_create makes of it on 256 CUs
        self._own_streams = []
        if overlap and self.reserved_cus > 0:
            ptr = C.c_void_p()
            _lib.check(_lib.load().ludwig_stream_create(device, self.reserved_cus, C.byref(ptr)))
            self._own_streams.append(ptr.value)
            self.s_comp = torch.cuda.ExternalStream(ptr.value, device=self.dev)
        else:
            self.s_comp = torch.cuda.current_stream(self.dev)
        self.level.set_stream(self.s_comp.cuda_stream)
        if order is not None:
            coords = np.asarray(view.level.active_block_coords)
            for part, mask in ((_lib.PART_ALL, np.ones(view.n_owned, bool)),
                               (_lib.PART_BOUNDARY, view.level.comm_boundary[: view.n_owned] != 0),
                               (_lib.PART_INTERIOR, view.level.comm_boundary[: view.n_owned] == 0)):
                ids = np.flatnonzero(mask)
                if ids.size:
                    items = order_mod.build(order, coords[ids])
                    items = np.where(items >= 0, (ids[np.maximum(items, 0) >> 3] << 3) | (items & 7), -1).astype(np.int32)
                    self.level.set_order(items, part)
        lib = _lib.load()
        handle = self.level.handle
        self.comm, self._own_comm = comm, False
        if self.transport == "native":
            if self.comm is None:
                try:
                    self.comm = native_comm(device)
                    self._own_comm = self.comm is not None
                except _lib.LudwigError as e:
                    # the library could not bring up its own RCCL communicator (no librccl to resolve, ncclCommInitRank refused):
                    # the same exchange over the RCCL communicator torch.distributed already holds - said loudly, never silently
                    import sys
                    print(f"[ludwig] native RCCL communicator unavailable ({e}); halo exchange falls back to torch.distributed", file=sys.stderr, flush=True)
                    self.transport = "torch"
        if self.transport == "native":
            self.ex = NativeHalo(plan, self.level, self.comm, wire_rank)
            self.s_comm = None
        else:
            def pack(name, idx, out):
                _lib.check(lib.ludwig_halo_pack(handle, _lib.FIELD_NAMES[name], C.c_void_p(idx.data_ptr()), idx.numel(),
                                                C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))

            def unpack(name, idx, src):
                _lib.check(lib.ludwig_halo_unpack(handle, _lib.FIELD_NAMES[name], C.c_void_p(idx.data_ptr()), idx.numel(),
                                                  C.c_void_p(src.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))

            hx = HaloExchanger(plan, view.rank, self.dev, pack, unpack, stage_through_host)
            if wire_rank:
                hx.wire_rank = dict(wire_rank)
            self.s_comm = torch.cuda.Stream(self.dev, priority=-1) if overlap else self.s_comp      # see init_rccl
            self.ex = TorchHalo(hx, self.s_comp, self.s_comm)

    def step(self, t: int, u_curr=0.0) -> None:
        """One step with the halo exchange hidden behind the NEXT step's interior blocks:
             compute stream : interior(t) | wait exchange(t-1) | boundary(t) | [f_post halo, Bouzidi correction(t)] |
             comm stream    :   exchange(t-1): pack, send/recv, unpack        |                                      exchange(t) ...
        Interior blocks (no ghost neighbour) read and write owned cells only, so they may run while the ghosts of their input are
        still arriving; the boundary blocks wait for them. Levels with Bouzidi cells follow the same schedule: the correction
        rewrites f_out after the collision from the post-collision values of neighbour cells, so its small f_post halo (only the
        links that reach across a cut) is exchanged in between and waited for, and the f / u halo goes last.
        Native transport: all of it is ONE call into the library (ludwig_step_distributed)."""
        from .physics import apply_bouzidi_correction, stream_collide
        _lib = self._lib
        out_f, out_v = ("f_temp", "vel_temp") if t % 2 == 0 else ("f", "vel")
        if not self.overlap:
            stream_collide(self.level, None, np.float32(0.5), u_curr, self.params, t, part=_lib.PART_ALL)
            if self.level.has_post_collision:
                if self.ex.plan.has("f_post"):
                    self.ex.post({"f_post": "f_post_collision"})
                    self.ex.join()
                apply_bouzidi_correction(self.level, t, self.params.q_min_threshold)
            self.ex.post({"f": out_f, "vel": out_v})
            self.ex.join()
            return
        if self.transport == "native":
            fl = self.params.to_c()
            _lib.check(_lib.load().ludwig_step_distributed(self.level.handle, self.ex.handle, None, int(t), float(np.float32(u_curr)), 0.5, 0.0,
                                                           self.C.byref(fl)))
            return
        stream_collide(self.level, None, np.float32(0.5), u_curr, self.params, t, part=_lib.PART_INTERIOR)
        self.ex.join()                                      # ghosts of this step's input are in place
        stream_collide(self.level, None, np.float32(0.5), u_curr, self.params, t, part=_lib.PART_BOUNDARY)
        if self.level.has_post_collision:
            if self.ex.plan.has("f_post"):
                self.ex.post({"f_post": "f_post_collision"})
                self.ex.join()
            apply_bouzidi_correction(self.level, t, self.params.q_min_threshold)
        self.ex.post({"f": out_f, "vel": out_v})

    def exchange_now(self, f_name: str, vel_name: str) -> None:
        """refresh the ghosts of the two named fields and wait (set-up: ghosts of a start state)"""
        self.ex.post({"f": f_name, "vel": vel_name})
        self.ex.join()

    def flush(self) -> None:
        """kept for callers of the round-2 schedule (the exchange used to be enqueued one launch late): nothing to do"""

    def synchronize(self) -> None:
        self.torch.cuda.synchronize(self.dev)

    def close(self, dist=None) -> None:
        """Teardown in a fixed order: (1) everything queued has run, (2) the halo plan and the library's communicator are destroyed,
        torch's wrappers of our streams and the events recorded on them dropped, (3) the process group is destroyed (pass
        torch.distributed as `dist` when this runner is the last user of the default group), (4) the level's device memory is
        freed, (5) the CU-masked stream underneath is destroyed. Round 2 destroyed the stream first, under live ExternalStream
        wrappers and a live communicator, and left the rest to interpreter shutdown: one profiled run ended in a SIGSEGV inside
        __cxa_finalize (profiles/README.md)."""
        import gc
        self.synchronize()
        self.ex.close()
        self.ex = None
        if self._own_comm:
            native_comm_destroy(self.comm)
        self.comm = None
        self.level.set_stream(None)
        self.s_comm = None
        self.s_comp = None
        gc.collect()
        if dist is not None and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        self.torch.cuda.synchronize(self.dev)
        self.level.close()
        for ptr in self._own_streams:
            self._lib.check(self._lib.load().ludwig_stream_destroy(self.dev.index, self.C.c_void_p(ptr)))
        self._own_streams = []


def distributed_level(global_level: BlockLevel, owner: np.ndarray, params, rank: int, world: int, device: int, overlap: bool = True,
                      stage_through_host: bool = False, transport: Optional[str] = None) -> "DistributedLevelRunner":
    """Partition a populated single-level case (any topology: tunnel with body, sponge, Bouzidi ...) by the block owner map."""
    view = build_local_level(global_level.level_id, global_level.active_block_coords, global_level.neighbor_table, owner, rank,
                             float(global_level.tau), temporal=global_level.f_old.size > 27)
    slice_level_fields(view, global_level)
    n_global = global_level.n_blocks
    mine = make_requests(view, n_global)
    to_me = exchange_requests(mine, world, rank) if world > 1 else {}
    plan = build_plan(view, n_global, mine, to_me)
    return DistributedLevelRunner(view, plan, params, device, overlap=overlap, stage_through_host=stage_through_host, transport=transport)


# ----------------------------------------------------------------------------------------------------------------
# multi-level (scope row N3): every level partitioned by the owner of its level-1 ancestor, parent-data ghosts
# ----------------------------------------------------------------------------------------------------------------
def ancestor_owner(level_id: int, coords, level1_coords, level1_owner: np.ndarray) -> np.ndarray:
    """owner of a level-l block = owner of the level-1 block that contains it (block coords shrink by 2 per level)"""
    lut = {tuple(c): int(o) for c, o in zip(np.asarray(level1_coords), level1_owner)}
    c = (np.asarray(coords, dtype=np.int64) - 1) >> (level_id - 1)
    return np.array([lut[(int(a) + 1, int(b) + 1, int(d) + 1)] for a, b, d in c], dtype=np.int64)


def bisect_owner(coords, weights, world: int) -> np.ndarray:
    """Recursive coordinate bisection of a block set with planar cuts between blocks: every cut takes the axis and plane
    that split the weight best (ties: the longer axis); rank ranges are halved alongside (any world size)."""
    c1 = np.asarray(coords, dtype=np.int64).reshape(-1, 3)
    w = np.asarray(weights, dtype=np.float64)
    owner = np.zeros(len(c1), dtype=np.int64)

    def split(ids: np.ndarray, r0: int, n: int) -> None:
        if n == 1 or ids.size == 0:
            owner[ids] = r0
            return
        n_lo = n // 2
        ext = c1[ids].max(axis=0) - c1[ids].min(axis=0)
        target = w[ids].sum() * n_lo / n
        best = None                            # (imbalance, -extent, axis, plane): the best balanced cut of the three axes
        for axis in range(3):
            planes = np.unique(c1[ids, axis])
            if planes.size < 2:
                continue
            below = np.array([w[ids][c1[ids, axis] <= p].sum() for p in planes[:-1]])
            j = int(np.argmin(np.abs(below - target)))
            cand = (float(abs(below[j] - target)), -int(ext[axis]), axis, planes[j])
            if best is None or cand[:2] < best[:2]:
                best = cand
        if best is None:                       # a single block left: nothing to cut
            owner[ids] = r0
            return
        lo = ids[c1[ids, best[2]] <= best[3]]
        hi = ids[c1[ids, best[2]] > best[3]]
        split(lo, r0, n_lo)
        split(hi, r0 + n_lo, n - n_lo)

    split(np.arange(len(c1)), 0, world)
    return owner


def balanced_owner(grids: Sequence[BlockLevel], world: int) -> np.ndarray:
    """Owner of every level-1 block when whole hierarchies stay on one rank (`ancestor_owner` for the finer levels):
    bisection weighted by the work below each level-1 block (its own step + 2^(l-1) sub-steps of every level-l descendant
    block per coarse step). Balance is limited by the level-1 block size; `level_owners` cuts every level on its own."""
    c1 = np.asarray(grids[0].active_block_coords, dtype=np.int64)
    lut = {tuple(c): i for i, c in enumerate(c1)}
    w = np.ones(len(c1), dtype=np.float64)
    for g in grids[1:]:
        anc = ((np.asarray(g.active_block_coords, dtype=np.int64).reshape(-1, 3) - 1) >> (g.level_id - 1)) + 1
        np.add.at(w, [lut[tuple(a)] for a in anc], float(2 ** (g.level_id - 1)))
    return bisect_owner(c1, w, world)


def level_owners(grids: Sequence[BlockLevel], world: int) -> List[np.ndarray]:
    """Every level cut on its own into `world` parts of equal block count (planar cuts at that level's block size). The
    levels are stepped one after the other with an exchange in between, so it is each level's balance that counts; the
    cuts of different levels need not line up - a fine block's parent may live on another rank (see required_parent_blocks)."""
    return [bisect_owner(g.active_block_coords, np.ones(g.n_blocks), world) for g in grids]


def required_parent_blocks(child: BlockLevel, child_owned: np.ndarray, parent: BlockLevel) -> np.ndarray:
    """Global ids (0-based) of the parent-level blocks the interface interpolation of the given owned child blocks can
    touch: for every owned child block with a missing neighbour, its parent block and that block's 26 neighbours (the 8
    stencil corners of a source cell just outside the child block lie within one parent cell of its parent's box)."""
    nt = np.asarray(child.neighbor_table)[child_owned]
    edge = child_owned[(nt == 0).any(axis=1)]
    if edge.size == 0:
        return np.zeros(0, dtype=np.int64)
    cc = np.asarray(child.active_block_coords, dtype=np.int64).reshape(-1, 3)[edge]
    pc = (cc + 1) // 2
    pptr = np.asarray(parent.block_pointer)
    inside = (pc >= 1).all(axis=1) & (pc[:, 0] <= pptr.shape[0]) & (pc[:, 1] <= pptr.shape[1]) & (pc[:, 2] <= pptr.shape[2])
    pid = np.zeros(len(pc), dtype=np.int64)
    pid[inside] = pptr[pc[inside, 0] - 1, pc[inside, 1] - 1, pc[inside, 2] - 1]
    pid = np.unique(pid[pid > 0]) - 1
    nb = np.asarray(parent.neighbor_table)[pid].reshape(-1)
    return np.unique(np.concatenate([pid, nb[nb > 0] - 1]))


def interpolation_needs(child: LocalView, parent: LocalView, domain_cells: Tuple[int, int, int]) -> Dict[str, np.ndarray]:
    """Parent-level ghost elements the child's coarse->fine interface reads (src/physics_interpolation.jl:29-62):
    for every link (owned child cell, k) whose source block is missing and whose source cell lies inside the global box,
    the 8 parent cells around the source position; f_k, rho and the 3 velocity components of those that live in parent
    GHOST blocks. Offsets are local to the parent level (reference layout)."""
    B = BLOCK_SIZE
    cl, pl = child.level, parent.level
    scale = 2 ** (cl.level_id - 1)
    nxg, nyg, nzg = (d * scale for d in domain_cells)
    table = np.asarray(cl.neighbor_table)[: child.n_owned]
    edge = np.flatnonzero((table == 0).any(axis=1))
    out = {"f": [], "rho": [], "vel": []}
    if edge.size == 0:
        return {k: np.zeros(0, np.int64) for k in out}
    bx = cl.map_x[edge].astype(np.int64); by = cl.map_y[edge].astype(np.int64); bz = cl.map_z[edge].astype(np.int64)
    pnb = pl.n_blocks
    pptr = np.asarray(pl.block_pointer)
    pdx, pdy, pdz = pptr.shape
    x, y, z = np.meshgrid(np.arange(B), np.arange(B), np.arange(B), indexing="ij")
    x, y, z = x.reshape(-1), y.reshape(-1), z.reshape(-1)
    for k in range(27):
        cx, cy, cz = int(_CX[k]), int(_CY[k]), int(_CZ[k])
        sx, sy, sz = x - cx, y - cy, z - cz
        ox = np.where(sx < 0, -1, np.where(sx >= B, 1, 0)); oy = np.where(sy < 0, -1, np.where(sy >= B, 1, 0)); oz = np.where(sz < 0, -1, np.where(sz >= B, 1, 0))
        cross = np.flatnonzero((ox != 0) | (oy != 0) | (oz != 0))
        if cross.size == 0:
            continue
        d = (ox + 1) + 3 * (oy + 1) + 9 * (oz + 1)
        missing = table[edge][:, d[cross]] == 0                                   # [n_edge, n_cross]
        gx = ((bx - 1) * B)[:, None] + x[cross][None, :] + 1 - cx                  # source cell, 1-based fine coords
        gy = ((by - 1) * B)[:, None] + y[cross][None, :] + 1 - cy
        gz = ((bz - 1) * B)[:, None] + z[cross][None, :] + 1 - cz
        inside = (gx >= 1) & (gx <= nxg) & (gy >= 1) & (gy <= nyg) & (gz >= 1) & (gz <= nzg)
        m = missing & inside
        if not m.any():
            continue
        sgx, sgy, sgz = gx[m], gy[m], gz[m]
        pc = [((s.astype(np.float32) - np.float32(0.5)) * np.float32(0.5)) for s in (sgx, sgy, sgz)]
        p0 = [np.floor(v).astype(np.int64) for v in pc]
        p1 = [v + 1 for v in p0]
        p0 = [np.maximum(v, 1) for v in p0]                                        # px1 before the clamp (Appendix A.14)
        for n in range(8):
            pg = [(p1[a] if (n >> a) & 1 else p0[a]) for a in range(3)]
            pb = [(v - 1) // B + 1 for v in pg]
            ok = (pb[0] >= 1) & (pb[0] <= pdx) & (pb[1] >= 1) & (pb[1] <= pdy) & (pb[2] >= 1) & (pb[2] <= pdz)
            idx = np.zeros(len(sgx), dtype=np.int64)
            idx[ok] = pptr[pb[0][ok] - 1, pb[1][ok] - 1, pb[2][ok] - 1]
            gh = idx > parent.n_owned                                              # lives in a parent ghost block
            if not gh.any():
                continue
            cell = ((pg[0][gh] - 1) % B) + B * ((pg[1][gh] - 1) % B) + B * B * ((pg[2][gh] - 1) % B)
            blk0 = idx[gh] - 1
            out["f"].append((k * pnb + blk0) * CELLS + cell)
            out["rho"].append(blk0 * CELLS + cell)
            for comp in range(3):
                out["vel"].append((comp * pnb + blk0) * CELLS + cell)
    return {k: (np.unique(np.concatenate(v)) if v else np.zeros(0, np.int64)) for k, v in out.items()}


class MultiLevelRunner:
    """Distributed recursive_step! (src/solver_control.jl:21-143) for nested levels: same call order and A/B parity as
    the single-device driver, plus after every level step the halo exchange of that level (same-level ghosts AND the
    parent-data ghosts its children interpolate from), overlapped with compute on a second HIP stream (_step_level).

    owners: one owner array per level (`level_owners`: every level cut on its own, the default of DistributedStepper), or a
    single level-1 array (whole hierarchies per rank: `balanced_owner` + `ancestor_owner`)."""

    def __init__(self, grids: Sequence[BlockLevel], owners, params, rank: int, world: int, device: int,
                 stage_through_host: bool = False, overlap: bool = True, transport: Optional[str] = None, comm=None,
                 wire_ranks: Optional[List[Dict[int, int]]] = None, requests_to_me: Optional[Callable] = None, upload_state: bool = True):
        """upload_state=False: the host levels' f / rho / vel arrays are neither sliced nor uploaded - the caller initialises the state
        on the device right away (DistributedStepper: init_eq!); on the shipped Wing_5_deg that is 25 GB of zeros per rank otherwise"""
        import ctypes as C
        import torch
        from . import _lib
        from .blocks import adapt
        self.params, self.rank, self.world = params, rank, world
        self.torch, self._lib = torch, _lib
        dims = (params.domain_nx, params.domain_ny, params.domain_nz)
        if isinstance(owners, np.ndarray):
            owner1 = owners
            owners = [owner1] + [ancestor_owner(g.level_id, g.active_block_coords, grids[0].active_block_coords, owner1) for g in grids[1:]]
        assert len(owners) == len(grids)
        # finest first: a level's ghost set includes the parent blocks this rank's finer blocks interpolate from
        views: List[Optional[LocalView]] = [None] * len(grids)
        for i in range(len(grids) - 1, -1, -1):
            g = grids[i]
            extra = None
            if i + 1 < len(grids):
                child_owned = np.flatnonzero(np.asarray(owners[i + 1]) == rank)
                extra = required_parent_blocks(grids[i + 1], child_owned, g)
            v = build_local_level(g.level_id, g.active_block_coords, g.neighbor_table, owners[i], rank, float(g.tau),
                                  temporal=g.f_old.size > 27, extra_ghosts=extra)
            slice_level_fields(v, g, state=upload_state)
            views[i] = v
        self.views = views
        torch.cuda.set_device(device)
        self.dev = torch.device("cuda", device)
        lib = _lib.load()
        # plans first: which blocks of a level send anything is part of the level's description (comm_boundary)
        plans: List[HaloPlan] = []
        for i, (v, g) in enumerate(zip(self.views, grids)):
            needs = compute_needs(v) if v.n_owned > 0 else {"f": np.zeros(0, np.int64), "vel": np.zeros(0, np.int64)}
            needs.setdefault("f_post", np.zeros(0, np.int64))
            needs.setdefault("rho", np.zeros(0, np.int64))
            if i + 1 < len(grids) and self.views[i + 1].n_owned > 0:
                extra = interpolation_needs(self.views[i + 1], v, dims)
                for name in ("f", "rho", "vel"):
                    needs[name] = np.unique(np.concatenate([needs[name], extra[name]]))
            mine = make_requests(v, g.n_blocks, needs)
            if requests_to_me is not None:          # loop-back tests: what the peers would ask, supplied by the caller (level index, own requests)
                to_me = requests_to_me(i, mine)
            else:
                to_me = exchange_requests(mine, world, rank) if world > 1 else {}
            plan = build_plan(v, g.n_blocks, mine, to_me)
            plans.append(plan)
            # "boundary" part = owned blocks next to a ghost block (they read ghosts) AND owned blocks any peer reads from (same-level
            # halo, or parent data of a peer's finer blocks - those may lie anywhere): stepped first, so that the exchange can run
            # under the remaining, interior blocks
            if v.n_owned > 0:
                sk = 512 * v.level.n_blocks
                for pr in plan.peers:
                    for name in FIELD_GROUPS:
                        off = plan.send[pr][name]
                        if len(off):
                            blk = (np.asarray(off, dtype=np.int64) % sk) // 512
                            v.level.comm_boundary[blk[blk < v.n_owned]] = 1
        # no local copy at all of a level: None (skipped); only ghost copies (parent data for finer blocks): kept, never stepped
        for v, plan in zip(self.views, plans):
            name_post_collision_readers(v.level, plan)
        self.levels = [adapt(v.level, device, upload_state) if v.level.n_blocks > 0 else None for v in self.views]
        self.plans = plans
        self.overlap = overlap
        self.transport = transport or ("torch" if stage_through_host else os.environ.get("LUDWIG_HALO_TRANSPORT", "native"))
        assert self.transport in ("native", "torch") and not (stage_through_host and self.transport == "native")
        self.s_comp = torch.cuda.current_stream(self.dev)
        for L in self.levels:
            if L is not None:
                L.set_stream(self.s_comp.cuda_stream)
        self.comm, self._own_comm = comm, False
        self.ex: List = []
        if self.transport == "native":
            # RCCL called from the library (NativeHalo): every level's plan has its own high-priority stream; one communicator
            if self.comm is None:
                try:
                    self.comm = native_comm(device)
                    self._own_comm = self.comm is not None
                except _lib.LudwigError as e:
                    import sys
                    print(f"[ludwig] native RCCL communicator unavailable ({e}); halo exchange falls back to torch.distributed", file=sys.stderr, flush=True)
                    self.transport = "torch"
        # per level: is hiding the exchange behind part of the level's own step worth two cross-stream hand-overs and two launches
        # instead of one? Only where a level step is long against them - nested levels take 2^(l-1) steps per coarse step, most of them
        # on levels of a few thousand blocks whose whole exchange is shorter than one hand-over (tools/multilevel_runner_host_cost.py).
        # Below LUDWIG_HALO_IN_STREAM_BELOW owned blocks (default 8 192 = 4.2 M cells, ~0.2 ms of stepping - the size below which the
        # library also merges a level's launches) the native exchange is queued on the level's own stream and the level is stepped in
        # one launch; a level this rank shares with nobody likewise. RCCL loop-back, two nested levels (profiles/
        # r03_nested_loopback_in_stream_ab.txt): 64 + 256 owned blocks 0.25 -> 0.17 ms per coarse step, 1 728 + 2 304 a wash.
        below = int(os.environ.get("LUDWIG_HALO_IN_STREAM_BELOW", "8192"))
        self.level_overlap = [bool(overlap)] * len(plans)
        if self.transport == "native":
            for i, plan in enumerate(plans):
                if self.levels[i] is None:
                    assert not plan.peers, "a level without a local copy exchanges nothing"
                    self.ex.append(None)
                else:
                    hx = NativeHalo(plan, self.levels[i], self.comm, wire_ranks[i] if wire_ranks else None)
                    if overlap and (not plan.peers or self.views[i].n_owned < below):
                        hx.set_in_stream(True)
                        self.level_overlap[i] = False
                    self.ex.append(hx)
            self.s_comm = None
        else:
            self.s_comm = torch.cuda.Stream(self.dev, priority=-1) if overlap else self.s_comp      # see init_rccl
            for i, plan in enumerate(plans):
                handle = self.levels[i].handle if self.levels[i] is not None else None

                def pack(name, idx, out, handle=handle):
                    _lib.check(lib.ludwig_halo_pack(handle, _lib.FIELD_NAMES[name], C.c_void_p(idx.data_ptr()), idx.numel(),
                                                    C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))

                def unpack(name, idx, src, handle=handle):
                    _lib.check(lib.ludwig_halo_unpack(handle, _lib.FIELD_NAMES[name], C.c_void_p(idx.data_ptr()), idx.numel(),
                                                      C.c_void_p(src.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))

                hx = HaloExchanger(plan, rank, self.dev, pack, unpack, stage_through_host)
                if wire_ranks:
                    hx.wire_rank = dict(wire_ranks[i])
                self.ex.append(TorchHalo(hx, self.s_comp, self.s_comm))

    def _join(self, i: int) -> None:
        """the compute stream waits for level i's exchange in flight (its ghosts are about to be read)"""
        if self.ex[i] is not None:
            self.ex[i].join()

    def _post(self, i: int, fields: Dict[str, str]) -> None:
        """level i's exchange, queued behind what the compute stream holds so far; what is launched afterwards runs under it"""
        if self.ex[i] is not None:
            self.ex[i].post(fields)

    def _post_and_join_now(self, i: int, fields: Dict[str, str]) -> None:
        """an exchange whose result is needed at once (the f_post halo before the Bouzidi correction): nothing can run under it, so the
        native transport queues it on the level's own stream - no hand-over to the plan's stream and back"""
        ex = self.ex[i]
        if ex is None:
            return
        if isinstance(ex, NativeHalo) and not ex.in_stream:
            ex.set_in_stream(True)           # (joins an exchange still in flight first)
            try:
                ex.post(fields)
            finally:
                ex.set_in_stream(False)
            return
        ex.post(fields)
        ex.join()

    def _step_level(self, i: int, t_sub: int, parent, parent_tau, tw, u, has_children: bool) -> None:
        """One level step + its halo exchange (same-level ghosts and the parent-data ghosts of peers' finer blocks).
        Overlap (the exchange runs on its own stream):
          * a level WITH children: boundary part (every block a ghost or a peer depends on) first, then the exchange under the
            interior part; the children start when both are done.
          * the finest level (nobody below reads its ghosts): interior part first - it reads no ghost - while the exchange of
            its PREVIOUS sub-step is still arriving, then the boundary part, [f_post halo, Bouzidi correction], and its own
            exchange is left in flight under whatever comes next (the next sub-step's interior part, or a coarser level's step)."""
        from .physics import apply_bouzidi_correction, stream_collide
        _lib = self._lib
        L, ex = self.levels[i], self.ex[i]
        stepping = self.views[i].n_owned > 0                 # else: only ghost copies here, refreshed by the exchange below
        out_f, out_v = ("f_temp", "vel_temp") if t_sub % 2 == 0 else ("f", "vel")
        fields = {"f": out_f, "vel": out_v}
        plan = ex.plan if ex is not None else None
        if plan is not None and plan.has("rho"):
            fields["rho"] = "rho"
        has_post_halo = plan is not None and plan.has("f_post")
        if i > 0:
            self._join(i - 1)                                # the parent's ghosts (interpolation stencils) must be in place
        if not self.level_overlap[i]:
            if stepping:
                stream_collide(L, parent, parent_tau, u, self.params, t_sub, tw, part=_lib.PART_ALL)
            if has_post_halo:
                self._post_and_join_now(i, {"f_post": "f_post_collision"})
            if stepping and L.has_post_collision:
                apply_bouzidi_correction(L, t_sub, self.params.q_min_threshold)
            self._post(i, fields)
            self._join(i)
            return
        bouzidi = stepping and L.has_post_collision
        if has_children and not bouzidi:
            self._join(i)
            if stepping:
                stream_collide(L, parent, parent_tau, u, self.params, t_sub, tw, part=_lib.PART_BOUNDARY)
            self._post(i, fields)                            # waits for the boundary part only: the interior part runs under it
            if stepping:
                stream_collide(L, parent, parent_tau, u, self.params, t_sub, tw, part=_lib.PART_INTERIOR)
            self._join(i)                                    # children interpolate from this level's ghosts next
            return
        if stepping:
            stream_collide(L, parent, parent_tau, u, self.params, t_sub, tw, part=_lib.PART_INTERIOR)
        self._join(i)
        if stepping:
            stream_collide(L, parent, parent_tau, u, self.params, t_sub, tw, part=_lib.PART_BOUNDARY)
        if has_post_halo:
            self._post_and_join_now(i, {"f_post": "f_post_collision"})
        if bouzidi:
            apply_bouzidi_correction(L, t_sub, self.params.q_min_threshold)
        self._post(i, fields)
        if has_children:
            self._join(i)

    def _rec(self, lvl: int, t_sub: int, parent, parent_tau, tw, u) -> None:
        if lvl > len(self.levels):
            return
        L = self.levels[lvl - 1]
        has_children = lvl < len(self.levels)
        if L is not None:
            if has_children and self.params.use_temporal_interp and L.has_temporal_storage:
                self._join(lvl - 1)
                L.copy_to_old(t_sub)
            self._step_level(lvl - 1, t_sub, parent, parent_tau, tw, u, has_children)
        if has_children:
            # the level's tau is a property of the level, known on every rank (also where it has no copy of it)
            tau = np.float32(self.views[lvl - 1].level.tau)
            self._rec(lvl + 1, 2 * t_sub, L, tau, np.float32(0.0), u)
            self._rec(lvl + 1, 2 * t_sub + 1, L, tau, np.float32(0.5), u)

    def step(self, t: int, u_curr=0.0) -> None:
        """one coarse step = recursive_step!(grids, 1, t, ...)"""
        self._rec(1, t, None, np.float32(0.5), np.float32(0.0), np.float32(u_curr))

    def synchronize(self) -> None:
        self.torch.cuda.synchronize(self.dev)

    def close(self, dist=None) -> None:
        """same order as DistributedLevelRunner.close: device idle, plans and communicator gone, process group gone, levels freed"""
        import gc
        if not self.levels and not self.ex:
            return
        self.synchronize()
        for ex in self.ex:
            if ex is not None:
                ex.close()
        self.ex = []
        if self._own_comm:
            native_comm_destroy(self.comm)
        self.comm = None
        self.s_comm = None
        gc.collect()
        if dist is not None and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        for lv in self.levels:
            if lv is not None:
                lv.close()
        self.levels = []


def weak_scaling_layout(world: int, nb: int) -> Tuple[Tuple[int, int, int], Tuple[int, int, int]]:
    """(blocks per rank along x, y, z; ranks along x, y, z) of bench.py's weak-scaling box: nb^3 blocks per rank whatever the
    world size. 2 and 4 ranks: cubic bricks, z and y cut (rank_grid). 8 ranks: the 2 nb-cube (BASELINE configs[3], 512^3 cells at
    nb = 32) cut 1 x 2 x 4 into bricks of 2 nb x nb x nb/2 blocks rather than 2 x 2 x 2 cubes - no x face at all (rank_grid's
    docstring; loop-back, same box: 0.875 against 0.920 ms per step) and at most 6.3 MB per peer and step; slabs (1 x 1 x 8) measure
    the same but put 12.6 MB on each of two links."""
    if world == 8 and nb % 2 == 0:
        return (2 * nb, nb, nb // 2), (1, 2, 4)
    return (nb, nb, nb), rank_grid(world)


def strong_scaling_layout(world: int, nb_global: int) -> Tuple[Tuple[int, int, int], Tuple[int, int, int]]:
    """(blocks per rank along x, y, z; ranks along x, y, z) for a FIXED global box of nb_global^3 blocks (bench.py --scaling strong;
    BASELINE configs[3]: the 512^3 box at 1 / 2 / 4 / 8 GPUs). Cuts as in weak_scaling_layout: z first, then y, never x while the
    bricks stay at least 4 blocks thick (an x face is single cells 32 B apart and breaks the runs of x-consecutive blocks):
    1 -> 1x1x1, 2 -> 1x1x2, 4 -> 1x2x2, 8 -> 1x2x4, 16 -> 1x4x4."""
    g = [1, 1, 1]
    w, a = world, 0
    while w > 1:
        if w % 2:
            raise ValueError("world size must be a power of two")
        g[2 - a % 2] *= 2                 # z, y, z, y ...
        w //= 2
        a += 1
    if any(nb_global % gi for gi in g) or min(nb_global // g[1], nb_global // g[2]) < 4:
        raise ValueError(f"a box of {nb_global}^3 blocks cannot be cut {g[0]}x{g[1]}x{g[2]}")
    return (nb_global // g[0], nb_global // g[1], nb_global // g[2]), tuple(g)


def periodic_box_plan(rank: int, world: int, nb_per_rank: Tuple[int, int, int], grid: Tuple[int, int, int], tau: float = 0.5006,
                      u0: float = 0.03, init: bool = True):
    """Host side of bench.py's N > 1 workload, no GPU needed: rank `rank`'s view of a periodic Taylor-Green box cut into `grid`
    bricks of nb_per_rank blocks, its halo plan (requests exchanged over the default process group) and the step parameters."""
    from . import cases
    from .physics import SolverParams
    grid = tuple(grid)
    assert grid[0] * grid[1] * grid[2] == world
    nbg = tuple(nb_per_rank[i] * grid[i] for i in range(3))
    coords, table, owner = periodic_box_topology(nbg, grid)
    widen = os.environ.get("LUDWIG_WIDEN_X_RUNS", "1") != "0"
    view = build_local_level(1, coords, table, owner, rank, tau, widen_x_runs=widen)
    if init:
        cases.init_taylor_green(view.level, tuple(8 * n for n in nbg), u0, share_ab_buffers=True)    # host level is only uploaded
    params = SolverParams(domain_nx=8 * nbg[0], domain_ny=8 * nbg[1], domain_nz=8 * nbg[2], wall_model_active=False, c_wale=0.5,
                          nu_sgs_bg=0.0005, inlet_turbulence=0.0, use_temporal_interp=False, sponge_blend_dist=False)
    n_global = len(coords)
    mine = make_requests(view, n_global)
    to_me = exchange_requests(mine, world, rank) if world > 1 else ({rank: mine[rank]} if rank in mine else {})
    plan = build_plan(view, n_global, mine, to_me)
    return view, plan, params, n_global


def periodic_weak_scaling_box(rank: int, world: int, nb_per_rank: Tuple[int, int, int], device: int, overlap: bool = True,
                              order: Optional[str] = None, stage_through_host: bool = False, tau: float = 0.5006, u0: float = 0.03,
                              grid: Optional[Tuple[int, int, int]] = None):
    """bench.py N > 1 workload: every rank owns an nb_per_rank brick of one periodic Taylor-Green box cut into `grid` ranks
    (default rank_grid(world); bench.py: weak_scaling_layout - 8 ranks x 32^3 blocks = BASELINE configs[3], 512^3 cells - or
    strong_scaling_layout for a fixed global box)."""
    grid = tuple(grid) if grid is not None else rank_grid(world)
    view, plan, params, n_global = periodic_box_plan(rank, world, nb_per_rank, grid, tau, u0)
    runner = DistributedLevelRunner(view, plan, params, device, overlap=overlap, stage_through_host=stage_through_host, order=order)
    runner.n_global_blocks = n_global
    return runner
