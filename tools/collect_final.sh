#!/bin/bash
# After `gpurun -- bash tools/final_profile.sh`: copy the capture that came back under gpurun_out/final into profiles/ (tracked).
set -e
R=${ROUND:-r03}
O=gpurun_out/final
cp $O/bench.json profiles/${R}_bench.json
cp $O/bench_under_rocprof.json profiles/${R}_bench_under_rocprofv3.json
cp $O/${R}_bench256_kernel_stats.txt $O/${R}_bench256_rocprofv3_kernel_stats.csv profiles/
cp $O/${R}_bench256_traffic.json profiles/${R}_bench256_traffic.json
cp $O/${R}_bench256_traffic.json profiles/traffic.json
cp $O/floor.txt profiles/${R}_bench256_same_box_floor.txt
cp $O/bench_driver_args.json profiles/${R}_bench_driver_args.json
cp $O/bench512.json profiles/${R}_bench512_single_gpu.json
cp $O/bench_strong_n1.json profiles/${R}_bench_strong_n1.json
cp $O/bench_reference_block_order.json profiles/${R}_bench_reference_block_order_same_box.json
cp $O/bench_2rank_rehearsal_weak.json profiles/${R}_bench_2rank_rehearsal_weak.json
cp $O/bench_2rank_rehearsal_strong.json profiles/${R}_bench_2rank_rehearsal_strong.json
cp $O/bench_again.json profiles/${R}_bench_again_same_box.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
from open_ludwig_amd import build
d = json.load(open('profiles/traffic.json'))
print('library digest', build.source_digest(), '| traffic.json', d['source_digest'], '| match', build.source_digest() == d['source_digest'])
for n in ['bench', 'bench_under_rocprof', 'bench_again', 'bench_driver_args', 'bench_reference_block_order', 'bench512', 'bench_strong_n1',
          'bench_2rank_rehearsal_weak', 'bench_2rank_rehearsal_strong']:
    b = json.loads(open('gpurun_out/final/%s.json' % n).read().strip().splitlines()[-1])
    r = b.get('roofline', {})
    print('%-30s %9.1f MLUPS %8.4f ms  frac %s  eager %s' % (n, b['value'], b['ms_per_step'], r.get('frac'), r.get('frac_eager_rho')))
PY
grep k_stream_collide_xrun profiles/${R}_bench256_kernel_stats.txt | head -2
head -4 profiles/${R}_bench256_same_box_floor.txt
