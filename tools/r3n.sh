#!/bin/bash
O=gpurun_out/r3n
mkdir -p $O
python - <<'PY'
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
a, b = ctypes.c_int(), ctypes.c_int()
print("hipDeviceGetStreamPriorityRange rc", hip.hipDeviceGetStreamPriorityRange(ctypes.byref(a), ctypes.byref(b)), "least", a.value, "greatest", b.value)
PY
for i in 1 2; do
for cfg in "0:" "2:" "2:LUDWIG_SIDE_STREAM_PRIORITY=-1" "2:LUDWIG_SIDE_STREAM_PRIORITY=1"; do
  m=${cfg%%:*}; extra=${cfg##*:}
  env LUDWIG_SPLIT_STEP=$m $extra python tools/case_speed.py re266k 1000 2>&1 | tail -1 | sed "s/^/sphere mode $m $extra: /"
  env LUDWIG_SPLIT_STEP=$m $extra python tools/case_speed.py wing 400 2>&1 | tail -1 | sed "s/^/wing   mode $m $extra: /"
done; done > $O/side_priority.txt 2>&1
cat $O/side_priority.txt
