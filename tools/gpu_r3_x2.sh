#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3x
mkdir -p $OUT
cd $R
A3D_BB_ITERS=5 A3D_BB_CPROFILE=4 python tools/backward_bench.py --step --reps 1 > $OUT/cprofile2.txt 2>&1
grep -E "training iteration" $OUT/cprofile2.txt
grep -A200 "was called by" $OUT/cprofile2.txt | cut -c1-200 | head -n 120
