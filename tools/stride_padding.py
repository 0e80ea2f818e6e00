"""Step time of the REAL stepping kernel against the distance between two populations in device memory.
The library pads the population stride internally (ludwig_level_population_stride); LUDWIG_STRIDE_PAD_BLOCKS=k overrides its rule,
so one host level is uploaded again and again with another k. usage: stride_padding.py [--nb NBX NBY NBZ] [--rule] [k ...]
  --rule : leave the choice to the library (no override) and print what it chose.
tools/stridebench.hip sweeps the same distance with a math-free emulation in seconds; this script confirms its picks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from open_ludwig_amd import adapt, cases
from open_ludwig_amd.physics import stream_collide

args = sys.argv[1:]
nb = (32, 32, 32)
if "--nb" in args:
    i = args.index("--nb")
    nb = tuple(int(v) for v in args[i + 1:i + 4])
    del args[i:i + 4]
rule = "--rule" in args
args = [a for a in args if a != "--rule"]
pads = [int(v) for v in args] or ([None] if rule else [0, 512, 2048, 2560, 0])
grids, params = cases.periodic_box(nb, upload_only=True)
n_blocks = grids[0].n_blocks
cells = n_blocks * 512
for k in pads:
    if k is None:
        os.environ.pop("LUDWIG_STRIDE_PAD_BLOCKS", None)
    else:
        os.environ["LUDWIG_STRIDE_PAD_BLOCKS"] = str(k)
    d = adapt(grids[0], 0)
    t = 1
    for _ in range(40):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(150):
        stream_collide(d, None, np.float32(0.5), np.float32(0.0), params, t); t += 1
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 150 * 1e3
    sk = d.population_stride()
    print(f"{n_blocks} blocks, pad {'rule' if k is None else k}: stride {sk // 512} blocks = {sk * 4 / 2**20:.3f} MiB: {ms:.4f} ms per step, "
          f"{cells / ms / 1e3:.0f} MLUPS", flush=True)
    d.close(); del d
