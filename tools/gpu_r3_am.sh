#!/bin/bash
# round 3, GPU session AM: occupancy bits in front of the level-0 grid (grid never cleared; empty cells cost a bit test),
# dense-kernel shape by row count.  Same-box A/B against the previous build (tools/bin/libbase.so).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/am
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_scene.py tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_backward.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
A3D_GRID=0 timeout 300 python -m pytest tests/test_gpu_scene.py tests/test_gpu_model.py -m gpu -x -q -k "scene or backbone" > $OUT/tests_hash.log 2>&1
echo "tests (hash level 0) rc=$?"; tail -2 $OUT/tests_hash.log
for lib in base new base new; do
  [ $lib = base ] && export A3D_LIB_PATH=$R/tools/bin/libbase.so || unset A3D_LIB_PATH
  echo "== $lib: 16 scenes"; LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "scene_|stem|k_dense|sum"
done
for lib in base new; do
  [ $lib = base ] && export A3D_LIB_PATH=$R/tools/bin/libbase.so || unset A3D_LIB_PATH
  echo "== $lib: one scene"; LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "scene_|stem|sum"
  echo "== $lib: four scenes"; LT_BATCH=4 python tools/layer_table.py 2>&1 | grep -E "scene_|stem|sum"
done
for lib in base new base new; do
  [ $lib = base ] && export A3D_LIB_PATH=$R/tools/bin/libbase.so || unset A3D_LIB_PATH
  echo "== bench steps-only $lib"
  python bench.py --steps-only --reps 7 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('latency_ms_per_scene'))"
done
