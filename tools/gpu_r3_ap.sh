#!/bin/bash
# round 3, GPU session AP: bounded search of the click simulator (A3D_CLICK_PRUNE=0 is the plain pass)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ap
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_clicks.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
for v in 0 1; do
  for cfg in "80000 0.05" "80000 0.3" "80000 0.6" "300000 0.3" "300000 0.6"; do
    echo "== A3D_CLICK_PRUNE=$v $cfg"; A3D_CLICK_PRUNE=$v python tools/click_bench.py $cfg 2>&1 | grep -v amdgpu.ids
  done
done
for v in 0 1; do
  echo "== training iterations A3D_CLICK_PRUNE=$v"
  A3D_CLICK_PRUNE=$v A3D_BB_ITERS=10 A3D_TRAIN_TIMING=1 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "train_one_step" | sed -e 's/, decoder forward.*//' | tail -9
  A3D_CLICK_PRUNE=$v A3D_BB_ITERS=10 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" | awk '{print $3, $4}' | tr '\n' ' '; echo
done
for v in 0 1; do
  echo "== bench A3D_CLICK_PRUNE=$v"
  A3D_CLICK_PRUNE=$v python bench.py --no-cpu-baseline --reps 5 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('eval_round_ms'), d.get('eval_rounds_per_s'))"
done
