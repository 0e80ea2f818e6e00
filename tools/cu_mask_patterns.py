"""Which CU-mask patterns cost the stepping kernel what (hipExtStreamCreateWithCUMask called directly). 256 CUs = 8 words."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from open_ludwig_amd import _lib, adapt, cases
from open_ludwig_amd.physics import stream_collide
hip = C.CDLL("libamdhip64.so")
grids, params = cases.periodic_box((32, 32, 32), upload_only=True)
d = adapt(grids[0], 0)
def mask_without(bits):
    m = [0xFFFFFFFF] * 8
    for b in bits: m[b // 32] &= ~(1 << (b % 32))
    return m
pats = {
    "null stream": None,
    "all 256 enabled": mask_without([]),
    "bits 0-7 off": mask_without(range(8)),
    "bits 0,32,..,224 off": mask_without(range(0, 256, 32)),
    "bits 33j off (library)": mask_without([33 * j for j in range(8)]),
    "bits 248-255 off": mask_without(range(248, 256)),
    "bits 224-255 off": mask_without(range(224, 256)),
    "bits 0,1 off": mask_without([0, 1]),
    "bit 0 off": mask_without([0]),
    # round 3: is the price the NUMBER of compute units or an imbalance between the shader engines of an XCD?
    "bits 0-31 off (CUs 0-3 of every XCD)": mask_without(range(32)),
    "bits 0-15 off (CUs 0-1 of every XCD)": mask_without(range(16)),
    "bits 192-255 off (CUs 24-31 of every XCD)": mask_without(range(192, 256)),
    "bits 0-7 + 64-71 + 128-135 + 192-199 off (CUs 0, 8, 16, 24)": mask_without([b + o for o in (0, 64, 128, 192) for b in range(8)]),
    "bits 0-7 + 8-15 off again as 16": mask_without(range(16)),
    "bits 224-255 off again": mask_without(range(224, 256)),
    "null stream again": None,
}
for name, m in pats.items():
    st = C.c_void_p()
    if m is not None:
        arr = (C.c_uint32 * 8)(*m)
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, arr)
        assert rc == 0, rc
        d.set_stream(st.value)
    else:
        d.set_stream(None)
    t = 1
    for _ in range(30):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 100 * 1e3
    print(f"{name:28s} {ms:.4f} ms per step", flush=True)
    d.set_stream(None)
    if m is not None: hip.hipStreamDestroy(st)
