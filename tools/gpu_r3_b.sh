#!/bin/bash
# round 3, GPU session B: k_conv_wl (weights resident in LDS) and the three-slot weight ring (A3D_SK_RING=3): parity + speed
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3b
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_conv.py -x -q > $OUT/pytest_default.txt 2>&1; tail -3 $OUT/pytest_default.txt
A3D_SK_RING=3 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -x -q > $OUT/pytest_ring3.txt 2>&1; tail -3 $OUT/pytest_ring3.txt
for V in 320000 80000; do
for CFG in "A3D_SK_RING=2 A3D_NO_WL=1" "A3D_SK_RING=2 A3D_NO_WL=0" "A3D_SK_RING=3 A3D_NO_WL=0"; do
  echo "== voxels $V $CFG" >> $OUT/conv_bench.txt
  env $CFG python tools/conv_bench.py --voxels $V --reps 15 2>&1 | grep -v amdgpu | grep -E "conv3|down|up_" >> $OUT/conv_bench.txt
done
done
cat $OUT/conv_bench.txt
for CFG in "A3D_SK_RING=2 A3D_NO_WL=1" "A3D_SK_RING=3 A3D_NO_WL=0"; do
  echo "== $CFG" >> $OUT/layers4.txt
  env $CFG LT_BATCH=4 python tools/layer_table.py 2>&1 | grep -v amdgpu >> $OUT/layers4.txt
  echo "== $CFG" >> $OUT/bench_quick.txt
  env $CFG python bench.py --steps 20 --warmup 5 --reps 7 --no-cpu-baseline --no-profile 2>&1 | grep '^{' >> $OUT/bench_quick.txt
done
grep -E "sum|==" $OUT/layers4.txt
python - <<PY
import json
for l in open("$OUT/bench_quick.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d.get("latency_ms_per_scene"))
    else: print(l.strip())
PY
