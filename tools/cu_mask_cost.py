"""Step time of the 256^3 periodic box (one launch per step) on streams from ludwig_stream_create(device, reserved): what leaving
compute units to the exchange costs the stepping kernel itself. usage: cu_mask_cost.py [reserved ...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from open_ludwig_amd import _lib, adapt, cases
from open_ludwig_amd.physics import stream_collide
res = [int(v) for v in sys.argv[1:]] or [0, 8, 16, 32, 0, 8]
grids, params = cases.periodic_box((32, 32, 32), upload_only=True)
d = adapt(grids[0], 0)
lib = _lib.load()
for r in res:
    ptr = C.c_void_p()
    _lib.check(lib.ludwig_stream_create(0, r, C.byref(ptr)))
    d.set_stream(ptr.value)
    t = 1
    for _ in range(30):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 100 * 1e3
    print(f"reserved {r:3d} CUs: {ms:.4f} ms per step", flush=True)
    d.set_stream(None)
    _lib.check(lib.ludwig_stream_destroy(0, ptr))
