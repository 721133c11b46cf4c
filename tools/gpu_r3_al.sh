#!/bin/bash
# round 3, GPU session AL: occupancy build of the dense kernel for the backbone's 1x1 convolutions (k_dense_occ) + shape sweep of k_dense
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/al
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_scene.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for v in "1 42" "0 42" "0 62" "0 121" "0 82" "1 42" "0 42"; do
  set -- $v
  echo "== A3D_DENSE_OCC=$1 A3D_DENSE_SHAPE=$2"
  A3D_DENSE_OCC=$1 A3D_DENSE_SHAPE=$2 LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "k_dense|sum"
done
echo "== one scene"
for v in 1 0; do A3D_DENSE_OCC=$v LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "k_dense|sum"; done
for v in 1 0 1 0; do
  echo "== bench steps-only A3D_DENSE_OCC=$v"
  A3D_DENSE_OCC=$v python bench.py --steps-only --reps 7 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"
done
