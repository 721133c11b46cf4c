#!/bin/bash
# round 3, GPU session W: training iterations with the batched click rounds; whole GPU suite
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3w
mkdir -p $OUT
cd $R
A3D_BB_ITERS=10 A3D_TRAIN_TIMING=1 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|train_one_step" > $OUT/train_a.txt
tail -n 22 $OUT/train_a.txt | cut -c1-260
A3D_BB_ITERS=10 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" > $OUT/train_c.txt
cat $OUT/train_c.txt | cut -c1-200
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 12 > $OUT/pytest.txt
cat $OUT/pytest.txt
