#!/bin/bash
# round 3, GPU session K: flash attention with persistent workgroups: tests, training phases, kernel trace
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3k
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_backward.py -x -q -k "flash or decoder_training_step or whole_network or reference_training" > $OUT/pytest.txt 2>&1; tail -n 5 $OUT/pytest.txt
A3D_BB_ITERS=8 A3D_TRAIN_TIMING=1 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|train_one_step" > $OUT/train_flash.txt
cat $OUT/train_flash.txt | cut -c1-220
bash tools/trace_train.sh r03 > $OUT/trace_train.log 2>&1; tail -n 36 $OUT/trace_train.log | cut -c1-150
