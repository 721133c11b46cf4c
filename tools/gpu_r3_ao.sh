#!/bin/bash
# round 3, GPU session AO: gathers two stages ahead (three fragment sets, DMA first / gathers last, vmcnt(NS) at the stage end) on the
# 96-column 3^3 layers: A3D_SK_DEEP=1 (low-register build, four workgroups per CU), =2 (step-wise build, three per CU)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ao
mkdir -p $OUT
cd $R
A3D_SK_DEEP=1 timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -x -q > $OUT/tests_deep1.log 2>&1
echo "tests (deep 1) rc=$?"; tail -3 $OUT/tests_deep1.log
A3D_SK_DEEP=2 timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -x -q > $OUT/tests_deep2.log 2>&1
echo "tests (deep 2) rc=$?"; tail -3 $OUT/tests_deep2.log
for v in 0 1 2 0 1 2; do
  echo "== A3D_SK_DEEP=$v: 16 scenes"
  A3D_SK_DEEP=$v LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "spconv< 96> conv3|sum"
done
for v in 0 1 2; do
  echo "== A3D_SK_DEEP=$v: one scene"
  A3D_SK_DEEP=$v LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "spconv< 96> conv3|sum"
done
for v in 0 1 2 0 1 2; do
  echo "== bench steps-only A3D_SK_DEEP=$v"
  A3D_SK_DEEP=$v python bench.py --steps-only --reps 7 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"
done
