#!/bin/bash
# deep-level conv layers (latency-bound): ablations and plan knobs, one layer at a time
OUT=gpurun_out/small_sweep.txt
: > $OUT
run() {  # label, env...
  echo "== $1" >> $OUT
  shift
  for V in 80000 320000; do
    for C in L2_conv3_64_64 L3_conv3_128_128 L4_conv3_256_256 L3_conv3_256_256 L2_conv3_128_128; do
      env "$@" python tools/conv_bench.py --voxels $V --reps 20 --only $C 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-60 | sed "s/^/  $V /" >> $OUT
    done
  done
}
run "base" A3D_X=0
run "dbg32 (launch only)" A3D_DBG=32
run "dbg16 (launch + ticket + search)" A3D_DBG=16
run "dbg15 (no loads, no MFMA, no hand-off)" A3D_DBG=15
run "dbg7 (no loads, no MFMA)" A3D_DBG=7
run "dbg8 (no hand-off)" A3D_DBG=8
cat $OUT
