#!/bin/bash
# round 3, GPU session AF: two-wave workgroups (32 rows per wave, six workgroups per CU) for the 96-column conv kernel
R=$GRAFT_REPO_ROOT
cd $R
A3D_SK_W2=1 timeout 1200 python -m pytest tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -n 4
timeout 600 python -m pytest tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -n 2
for W in 0 1; do
  echo "== A3D_SK_W2=$W"
  A3D_SK_W2=$W LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "spconv< 96>" | head -n 12
  A3D_SK_W2=$W LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "spconv< 96>|sum" | tail -n 4
  A3D_SK_W2=$W python bench.py --steps 20 --warmup 5 --reps 7 --no-cpu-baseline --steps-only 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['frac'],4))"
done
