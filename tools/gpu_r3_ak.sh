#!/bin/bash
# round 3, GPU session AK: eight-wave workgroups (128-row tiles) for the 96-column 3^3 layers (A3D_SK_NW8 = minimum rows)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ak
mkdir -p $OUT
cd $R
A3D_SK_NW8=1 timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -x -q > $OUT/tests_nw8.log 2>&1
echo "tests (forced eight-wave) rc=$?"; tail -3 $OUT/tests_nw8.log
for v in 0 100000; do
  echo "== A3D_SK_NW8=$v"
  A3D_SK_NW8=$v LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "spconv< 96> conv3|sum" | tee $OUT/lt16_nw8_$v.txt
done
for v in 0 10000; do
  echo "== one scene A3D_SK_NW8=$v"
  A3D_SK_NW8=$v LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "spconv< 96> conv3|sum" | tee $OUT/lt1_nw8_$v.txt
done
for v in 0 100000 0 100000; do
  echo "== bench steps-only A3D_SK_NW8=$v"
  A3D_SK_NW8=$v python bench.py --steps-only --reps 7 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"
done
