"""prints the table of tools/rccl_loopback_table.sh"""
import glob, json, os, sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3t"
for f in ("single", "single2"):
    d = json.load(open(f"{O}/{f}.json")); print(f"single device bench.py: {d['ms_per_step']} ms per step, {d['value']} MLUPS")
for f in sorted(glob.glob(f"{O}/loop_*.json"), key=os.path.getmtime):
    d = json.load(open(f))
    print(f"{os.path.basename(f)[5:-5]:24s} {d.get('transport', '?'):6s} {d['peers']} peers, {d['halo_bytes_per_step'] / 1e6:.1f} MB per step, blocks in the view {d.get('view_blocks')}: "
          f"{d['ms_per_step_wall']:.4f} ms per step, exchange span {d['exchange_ms_median_after_first']:.3f} ms, host {d.get('host_us_per_step_enqueue', float('nan')):.0f} us per step to enqueue")
