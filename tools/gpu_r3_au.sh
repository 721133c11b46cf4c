#!/bin/bash
# round 3, GPU session AU: per-scene cache of the first decoder layer's click-to-scene keys / values (A3D_KV_CACHE_MB=0: off)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/au
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_clicks.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log
for v in 0 4096 0 4096; do
  echo "== A3D_KV_CACHE_MB=$v"
  A3D_KV_CACHE_MB=$v python bench.py --no-cpu-baseline --reps 5 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('decoder_pass_ms_single'), d.get('eval_round_ms'), d.get('eval_rounds_per_s'), d.get('iou_at_k',{}).get('max_abs_diff'))"
done
for v in 0 4096; do
  echo "== training A3D_KV_CACHE_MB=$v"
  A3D_KV_CACHE_MB=$v A3D_BB_ITERS=10 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" | awk '{print $3, $4}' | tr '\n' ' '; echo
done
