#!/bin/bash
# round 3, GPU session AT: IoU counts + error clusters of a round behind one host synchronisation
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/at
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_clicks.py tests/test_datasets.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for i in 1 2; do
  python bench.py --no-cpu-baseline --reps 5 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('eval_round_ms'), d.get('eval_rounds_per_s'), d.get('iou_at_k',{}).get('max_abs_diff'))"
done
python bench.py --voxels 300000 --clicks-per-object 4 --batch 1 --streams 2 --steps 10 --warmup 3 --reps 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5', d['value'], d.get('eval_round_ms'), d.get('eval_rounds_per_s'))"
