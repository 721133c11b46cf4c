#!/bin/bash
# round 3, GPU session AS: decoder groups of a heterogeneous batch on side streams (A3D_DEC_SIDE=0: all on the caller's stream)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/as
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_clicks.py tests/test_gpu_backward.py tests/test_gpu_distributed.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for v in 0 1 0 1; do
  echo "== training iterations A3D_DEC_SIDE=$v"
  A3D_DEC_SIDE=$v A3D_BB_ITERS=10 A3D_TRAIN_TIMING=1 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "train_one_step" | sed -e 's/, decoder forward.*//' -e 's/train_one_step: backbone forward [0-9.]* ms, //' | tail -9 | tr '\n' ';'; echo
  A3D_DEC_SIDE=$v A3D_BB_ITERS=10 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" | awk '{print $3, $4}' | tr '\n' ' '; echo
done
for v in 0 1; do
  echo "== bench A3D_DEC_SIDE=$v"
  A3D_DEC_SIDE=$v python bench.py --no-cpu-baseline --reps 5 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('eval_round_ms'), d.get('eval_rounds_per_s'), d.get('iou_at_k',{}).get('max_abs_diff'))"
done
