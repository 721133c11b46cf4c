#!/bin/bash
# Scratch script of the current GPU session (overwritten per session; `gpurun -- 'bash tools/gpu_session.sh'`).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s7
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_scene.py -x -q > $O/t_new.log 2>&1; echo "rc=$?" >> $O/t_new.log
timeout 2700 python -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
LT_BATCH=16 timeout 300 python tools/layer_table.py > $O/layer_table_16scenes.txt 2>&1
LT_BATCH=1 timeout 300 python tools/layer_table.py > $O/layer_table_1scene.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err

cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trt
A3D_BB_ITERS=6 A3D_TRAIN_TIMING=mark timeout 900 rocprofv3 --kernel-trace -d /tmp/trt -o t -- python $R/tools/backward_bench.py --step --reps 1 > $O/train_trace.log 2>&1
python $R/tools/train_phase_trace.py /tmp/trt 2 40 > $O/train_phases.txt 2>&1
ls -la $O
