#!/bin/bash
# round 3, GPU session G: scenes per step x steps in flight
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3g
mkdir -p $OUT
cd $R
for B in 4 8 16; do for S in 2 3 4 6; do
  echo "== batch $B streams $S" >> $OUT/sweep.txt
  python bench.py --batch $B --streams $S --steps $((80 / B)) --warmup 3 --reps 7 --steps-only --no-profile 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))" >> $OUT/sweep.txt
done; done
cat $OUT/sweep.txt
