"""Where a rank's step of the weak-scaling layouts goes before any exchange: the brick VIEW of tests/_rccl_loopback_worker.py (owned blocks +
ghost copies) stepped as one launch / as interior + boundary launches, on an ordinary stream and on the CU-masked stream the multi-GPU
schedule uses, against the plain box of the same cells. No exchange, no ghost refresh (timing only).
usage: split_penalty_view.py [grid=1x1x2] [nb=32 | nbx,nby,nbz]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from open_ludwig_amd import _lib, adapt, cases
from open_ludwig_amd.physics import stream_collide
import _rccl_loopback_worker as lw
grid = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1x1x2").split("x"))
nb = sys.argv[2] if len(sys.argv) > 2 else "32"
nb = int(nb) if "," not in nb else tuple(int(v) for v in nb.split(","))
nb3 = (nb, nb, nb) if isinstance(nb, int) else nb
view, plan, vparams = lw.symmetric_brick_plan(grid, nb)
plain, pparams = cases.periodic_box(nb3, upload_only=True)
lib = _lib.load()

def timed(d, params, parts, n=120):
    t = 1
    for _ in range(30):
        for p in parts: stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t, part=p)
        t += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for p in parts: stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t, part=p)
        t += 1
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for rep in range(2):
    d = adapt(plain[0], 0)
    print(f"plain box {nb3}: one launch {timed(d, pparams, (_lib.PART_ALL,)):.4f} ms", flush=True)
    d.close()
    for cus in (0, 8):
        d = adapt(view.level, 0)
        ptr = C.c_void_p()
        _lib.check(lib.ludwig_stream_create(0, cus, C.byref(ptr)))
        d.set_stream(ptr.value)
        nbnd = int((view.level.comm_boundary[: view.n_owned] != 0).sum())
        a = timed(d, vparams, (_lib.PART_ALL,)); b = timed(d, vparams, (_lib.PART_INTERIOR, _lib.PART_BOUNDARY))
        io = timed(d, vparams, (_lib.PART_INTERIOR,)); bo = timed(d, vparams, (_lib.PART_BOUNDARY,))
        print(f"  view {grid} ({view.n_owned} owned + {view.level.n_blocks - view.n_owned} ghost blocks, boundary part {nbnd}), {cus} CUs reserved: one launch {a:.4f} | "
              f"interior + boundary {b:.4f} | alone: interior {io:.4f}, boundary {bo:.4f}", flush=True)
        d.set_stream(None); d.close()
        _lib.check(lib.ludwig_stream_destroy(0, ptr))
