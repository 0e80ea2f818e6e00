"""What stepping a 256^3 periodic box as interior part + boundary part costs against one launch, with no exchange at all.
The boundary part is the one-block shell of the box (what a 2x2x2 brick decomposition marks). usage: split_penalty.py [blocks per edge]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from open_ludwig_amd import _lib, adapt, cases
from open_ludwig_amd.physics import stream_collide
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32
grids, params = cases.periodic_box((nb, nb, nb), upload_only=True)
g = grids[0]
c = np.asarray(g.active_block_coords)
shell = ((c == 1) | (c == nb)).any(axis=1)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _rccl_loopback_worker as lw
view, plan, vparams = lw.symmetric_brick_plan((2, 2, 2), nb)      # rank 0's brick of a 2x2x2 decomposition: the same blocks + ghost copies
configs = [("shell", shell, g, params), ("shell widened to aligned groups of 4 in x", shell | np.isin((c[:, 0] - 1) // 4, (0, (nb - 1) // 4)), g, params),
           ("brick view: owned blocks + ghost blocks behind them (widened)", None, view.level, vparams)]
for name, mask, g, params in configs:
    if mask is None:
        mask = g.comm_boundary[: view.n_owned] != 0
    else:
        g.comm_boundary = mask.astype(np.uint8)
    d = adapt(g, 0)
    st = torch.cuda.current_stream()
    d.set_stream(st.cuda_stream)
    def run(parts, n=100):
        t = 1
        for _ in range(20):
            for p in parts: stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t, part=p)
            t += 1
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            for p in parts: stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t, part=p)
            t += 1
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    s2 = torch.cuda.Stream()
    ev_i, ev_b = [torch.cuda.Event(), torch.cuda.Event()], torch.cuda.Event()
    def run_two_streams(n=100):
        """interior on one stream, boundary on another: each waits for the other part of the PREVIOUS step only"""
        t = 1
        def step(t, k):
            if k: st.wait_event(ev_b)
            d.set_stream(st.cuda_stream); stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t, part=_lib.PART_INTERIOR)
            ev_i[k % 2].record(st)
            if k: s2.wait_event(ev_i[(k - 1) % 2])
            d.set_stream(s2.cuda_stream); stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t, part=_lib.PART_BOUNDARY)
            ev_b.record(s2)
            d.set_stream(st.cuda_stream)
        k = 0
        for _ in range(20):
            step(t, k); t += 1; k += 1
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            step(t, k); t += 1; k += 1
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(f"{name}: boundary part {int(mask.sum())} of {len(mask)} blocks", flush=True)
    for rep in range(2):
        a = run((_lib.PART_ALL,)); b = run((_lib.PART_INTERIOR, _lib.PART_BOUNDARY))
        bo = run((_lib.PART_BOUNDARY,)); io = run((_lib.PART_INTERIOR,))     # one part over and over: timing only
        two = run_two_streams()
        print(f"  one launch {a:.4f} ms | interior + boundary {b:.4f} ms, on two streams {two:.4f} ms | alone: interior {io:.4f}, boundary {bo:.4f}", flush=True)
    d.close()
