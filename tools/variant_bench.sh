#!/bin/bash
# time library variants with bench.py (separate processes, same device, interleaved rounds)
for r in 1 2 3; do
for v in "$@"; do
  LUDWIG_HIP_LIB=$PWD/tools/variants/$v.so python bench.py --steps 100 --warmup 10 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'r$r', d['roofline']['kernel_ms'], d['value'])"
done
done
