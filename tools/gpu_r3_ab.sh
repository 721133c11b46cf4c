#!/bin/bash
# round 3, GPU session AB: do persistent grids that leave a few workgroup slots free let other streams' small kernels through?
R=$GRAFT_REPO_ROOT
cd $R
for CFG in "A3D_SK_FREE=0 A3D_DEC_FREE=0" "A3D_SK_FREE=32 A3D_DEC_FREE=0" "A3D_SK_FREE=64 A3D_DEC_FREE=0" "A3D_SK_FREE=0 A3D_DEC_FREE=8" "A3D_SK_FREE=32 A3D_DEC_FREE=8" "A3D_SK_FREE=64 A3D_DEC_FREE=16" "A3D_SK_FREE=128 A3D_DEC_FREE=16"; do
  echo "== $CFG"
  env $CFG python bench.py --steps 20 --warmup 5 --reps 7 --no-cpu-baseline --steps-only > /tmp/b.json 2> /tmp/b.err
  tail -n 3 /tmp/b.err
  python -c "
import sys, json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
done
