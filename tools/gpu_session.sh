#!/bin/bash
# s27: click simulator: cell-list near field + bounded brute force for the rest (A3D_CLICK_PRUNE=3 = bounded search alone)
mkdir -p gpurun_out/s27
timeout 1500 python -m pytest tests/test_gpu_clicks.py -x -q > gpurun_out/s27/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s27/pytest.log
{
for spec in "80000 0.02" "80000 0.05" "80000 0.3" "80000 0.6" "80000 0.9" "300000 0.1" "300000 0.3" "300000 0.6"; do
  for p in 3 1 3 1; do echo -n "PRUNE=$p $spec: "; A3D_CLICK_PRUNE=$p timeout 300 python tools/click_bench.py $spec 2>&1 | grep -E "get_simulated|voxels" | tr '\n' ' '; echo; done
done
} > gpurun_out/s27/click_bench.log 2>&1
for p in 3 1 3 1; do echo "== PRUNE=$p"; A3D_CLICK_PRUNE=$p A3D_TRAIN_TIMING=1 A3D_BB_ITERS=10 timeout 600 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "^train_one_step" | sed 's/backbone forward [0-9.]* ms, //; s/, losses.*//'; done > gpurun_out/s27/train.log 2>&1
A3D_BB_ITERS=4 bash tools/trace_train.sh s27 > /dev/null 2>&1; mv gpurun_out/s27_training_kernel_trace.txt gpurun_out/s27/ 2>/dev/null
