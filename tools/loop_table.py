"""prints the table of tools/rccl_loopback_table.sh from gpurun_out/r2t"""
import glob, json, os
for f in ("single", "single2"):
    d = json.load(open(f"gpurun_out/r2t/{f}.json")); print(f"single device bench.py: {d['ms_per_step']} ms per step, {d['value']} MLUPS")
for f in sorted(glob.glob("gpurun_out/r2t/loop_*.json"), key=os.path.getmtime):
    d = json.load(open(f))
    print(f"{os.path.basename(f)[5:-5]:28s} {d['peers']} peers, {d['halo_bytes_per_step'] / 1e6:.1f} MB per step, blocks in the view {d.get('view_blocks')}: {d['ms_per_step_wall']:.4f} ms per step, exchange span {d['exchange_ms_median_after_first']:.3f} ms")
