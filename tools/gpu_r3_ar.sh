#!/bin/bash
# round 3, GPU session AR: wave-aggregated atomics in k_cluster_best on top of the bounded click search
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ar
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_clicks.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for v in 1 0; do
  for cfg in "80000 0.05" "80000 0.3" "80000 0.6" "300000 0.1" "300000 0.3" "300000 0.6"; do
    echo "== A3D_CLICK_PRUNE=$v: $cfg"; A3D_CLICK_PRUNE=$v python tools/click_bench.py $cfg 2>&1 | grep median
  done
done
echo "== training iterations"
A3D_BB_ITERS=10 A3D_TRAIN_TIMING=1 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "train_one_step" | sed -e 's/, decoder forward.*//' | tail -9
A3D_BB_ITERS=10 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" | awk '{print $3, $4}' | tr '\n' ' '; echo
python bench.py --no-cpu-baseline --reps 5 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('eval_round_ms'), d.get('eval_rounds_per_s'))"
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_distributed.py -m gpu -x -q 2>&1 | tail -3
