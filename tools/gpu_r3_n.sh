#!/bin/bash
# round 3, GPU session N: batched decoder tape: tests + training iteration phases (cycle collector held off)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3n
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_backward.py tests/test_gpu_distributed.py -x -q > $OUT/pytest.txt 2>&1; tail -n 12 $OUT/pytest.txt
A3D_BB_ITERS=10 A3D_TRAIN_TIMING=1 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|train_one_step" > $OUT/train_a.txt
cat $OUT/train_a.txt | cut -c1-230
echo "== untimed (no phase syncs)"
A3D_BB_ITERS=10 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" > $OUT/train_c.txt
cat $OUT/train_c.txt | cut -c1-200
bash tools/trace_train.sh r03 > $OUT/trace_train.log 2>&1; tail -n 44 $OUT/trace_train.log | cut -c1-150 | head -30
