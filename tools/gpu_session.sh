#!/bin/bash
# s25: click simulator with a coarse bounding stage in front (A3D_CLICK_PRUNE=3 = the one-stage search)
mkdir -p gpurun_out/s25
timeout 1500 python -m pytest tests/test_gpu_clicks.py -x -q > gpurun_out/s25/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s25/pytest.log
{
for spec in "80000 0.05" "80000 0.3" "80000 0.6" "80000 0.9" "300000 0.1" "300000 0.3" "300000 0.6"; do
  for p in 3 1 3 1; do echo -n "PRUNE=$p $spec: "; A3D_CLICK_PRUNE=$p timeout 300 python tools/click_bench.py $spec 2>&1 | grep -E "get_simulated|voxels" | tr '\n' ' '; echo; done
done
} > gpurun_out/s25/click_bench.log 2>&1
for p in 3 1 3 1; do echo "== PRUNE=$p"; A3D_CLICK_PRUNE=$p A3D_TRAIN_TIMING=1 A3D_BB_ITERS=10 timeout 600 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "^train_one_step" | sed 's/backbone forward [0-9.]* ms, //; s/, decoder forward.*//'; done > gpurun_out/s25/train.log 2>&1
A3D_BB_ITERS=4 bash tools/trace_train.sh s25 > /dev/null 2>&1; mv gpurun_out/s25_training_kernel_trace.txt gpurun_out/s25/ 2>/dev/null
