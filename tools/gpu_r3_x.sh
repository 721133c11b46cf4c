#!/bin/bash
# round 3, GPU session X: one-launch weight repack; host profile of a training iteration with many click rounds
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3x
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -x -q -k "trains_with_clip or training_step or one_step or epoch" 2>&1 | tail -n 6
A3D_BB_ITERS=5 A3D_BB_CPROFILE=4 python tools/backward_bench.py --step --reps 1 > $OUT/cprofile.txt 2>&1
grep -E "training iteration" $OUT/cprofile.txt
grep -A60 "cumulative" $OUT/cprofile.txt | cut -c1-170 | head -n 64
