#!/bin/bash
# round 3, GPU session AG: five workgroups per CU (96 VGPRs) for the 96-column conv kernel -- experimental second library
R=$GRAFT_REPO_ROOT
cd $R
export A3D_LIB_PATH=$R/agile3d_amd/libagile3d_hip_occ5.so
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "conv3 or determin" 2>&1 | tail -n 3
for G in 1024 1280; do
  echo "== occ5 library, A3D_SK_G=$G"
  A3D_SK_G=$G LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "spconv< 96>" | sed -n 5,10p
  A3D_SK_G=$G python bench.py --steps 20 --warmup 5 --reps 7 --no-cpu-baseline --steps-only 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['frac'],4))"
done
