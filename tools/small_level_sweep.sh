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
run base A3D_X=0
run "dbg1 (no gather)" A3D_DBG=1
run "dbg2 (no weight DMA)" A3D_DBG=2
run "dbg3 (no loads)" A3D_DBG=3
run "dbg7 (no loads, no MFMA)" A3D_DBG=7
run minshare2 A3D_SK_MINSHARE=2
run minshare4 A3D_SK_MINSHARE=4
run minshare12 A3D_SK_MINSHARE=12
run minshare24 A3D_SK_MINSHARE=24
run smallch128 A3D_SK_SMALLCH=128
run "smallch128 minshare3" A3D_SK_SMALLCH=128 A3D_SK_MINSHARE=3
cat $OUT
