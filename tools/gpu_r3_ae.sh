#!/bin/bash
# round 3, GPU session AE: k_s2c_out with twelve waves per workgroup (three per SIMD, no register prefetch)
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -n 4
for W in 12 8; do
  echo "== A3D_S2C_WAVES=$W"
  A3D_S2C_WAVES=$W LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "s2c_attn|c2s_attn" | tail -n 4
  A3D_S2C_WAVES=$W LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "s2c_attn|sum" | tail -n 3
  A3D_S2C_WAVES=$W python bench.py --steps 20 --warmup 5 --reps 7 --no-cpu-baseline --steps-only 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
done
