import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from open_ludwig_amd import adapt, cases
from open_ludwig_amd.physics import stream_collide
grids, params = cases.periodic_box((32, 32, 32), upload_only=True)
level = adapt(grids[0], 0)
stream = torch.cuda.current_stream()
level.set_stream(stream.cuda_stream)
torch.cuda.synchronize()
n = 80
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for i in range(n):
    ev[i][0].record(stream); stream_collide(level, None, np.float32(0.5), np.float32(0.0), params, i + 1); ev[i][1].record(stream)
torch.cuda.synchronize()
t = [a.elapsed_time(b) for a, b in ev]
print("per-step ms, steps 1..80 right after the upload:")
print(" ".join(f"{x:.3f}" for x in t))
time.sleep(2.0)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for i in range(30):
    ev[i][0].record(stream); stream_collide(level, None, np.float32(0.5), np.float32(0.0), params, n + i + 1); ev[i][1].record(stream)
torch.cuda.synchronize()
print("after 2 s of idling:")
print(" ".join(f"{a.elapsed_time(b):.3f}" for a, b in ev))
