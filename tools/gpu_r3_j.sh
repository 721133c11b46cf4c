#!/bin/bash
# round 3, GPU session J: training iteration phases with / without the flash attention; kernel trace of four iterations;
# scenes per step x steps in flight with the full protocol
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3j
mkdir -p $OUT
cd $R
A3D_BB_ITERS=8 A3D_TRAIN_TIMING=1 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|train_one_step" > $OUT/train_flash.txt
A3D_BB_ITERS=8 A3D_TRAIN_TIMING=1 A3D_TRAIN_FLASH=0 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|train_one_step" > $OUT/train_noflash.txt
echo "== flash"; cat $OUT/train_flash.txt; echo "== materialised"; cat $OUT/train_noflash.txt
bash tools/trace_train.sh r03 > $OUT/trace_train.log 2>&1; tail -n 50 $OUT/trace_train.log | cut -c1-150
for CFG in "8 4" "16 3" "16 4" "16 6"; do set -- $CFG
  echo "== batch $1 streams $2" >> $OUT/sweep.txt
  python bench.py --batch $1 --streams $2 --steps 20 --warmup 3 --reps 9 --steps-only --no-profile 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), d['timed_region']['ms_per_step_all'])" >> $OUT/sweep.txt
done
cat $OUT/sweep.txt
