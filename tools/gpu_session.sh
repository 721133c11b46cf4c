#!/bin/bash
# Scratch script of the current GPU session (overwritten per session; `gpurun -- 'bash tools/gpu_session.sh'`).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s3
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_backward.py -x -q > $O/t_backward.log 2>&1; echo "rc=$?" >> $O/t_backward.log
timeout 1500 python -m pytest tests/test_gpu_distributed.py -x -q > $O/t_dist.log 2>&1; echo "rc=$?" >> $O/t_dist.log
timeout 1500 python -m pytest tests/test_gpu_fit.py -x -q -s > $O/t_fit.log 2>&1; echo "rc=$?" >> $O/t_fit.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trt
A3D_BB_ITERS=6 A3D_TRAIN_TIMING=mark timeout 900 rocprofv3 --kernel-trace -d /tmp/trt -o t -- python $R/tools/backward_bench.py --step --reps 1 > $O/train_trace.log 2>&1
python $R/tools/train_phase_trace.py /tmp/trt 2 40 > $O/train_phases.txt 2>&1
cd $R
A3D_BB_ITERS=8 A3D_TRAIN_TIMING=1 timeout 600 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|train_one_step" > $O/training_iterations.txt
timeout 2700 python -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
ls -la $O
