#!/bin/bash
# round 3, GPU session AD: larger batches
R=$GRAFT_REPO_ROOT
cd $R
for CFG in "16 4 20" "32 4 10" "32 3 10" "64 2 6" "64 3 6"; do
  set -- $CFG
  echo "== batch $1 streams $2"
  python bench.py --batch $1 --streams $2 --steps $3 --warmup 3 --reps 7 --no-cpu-baseline --steps-only > /tmp/b.json 2> /tmp/b.err
  tail -n 2 /tmp/b.err | grep -v amdgpu.ids
  python -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['frac'])"
done
