#!/bin/bash
# round 3, GPU session L: training iteration: device-side mask (no host syncs), allocator behaviour; decoder gradient tests
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3l
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_backward.py -x -q -k "decoder_training_step or whole_network or reference_training or full_training" > $OUT/pytest.txt 2>&1; tail -n 4 $OUT/pytest.txt
A3D_BB_ITERS=10 A3D_TRAIN_TIMING=1 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|train_one_step" > $OUT/train_a.txt
cat $OUT/train_a.txt | cut -c1-230
echo "== expandable segments"
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True A3D_BB_ITERS=10 A3D_TRAIN_TIMING=1 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|train_one_step|arn" > $OUT/train_b.txt
cat $OUT/train_b.txt | cut -c1-230
echo "== untimed (no phase syncs)"
A3D_BB_ITERS=10 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" > $OUT/train_c.txt
cat $OUT/train_c.txt | cut -c1-200
