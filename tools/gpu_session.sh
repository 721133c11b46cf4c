#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
rocm-smi --showpower --showclocks --showmaxpower --showtemp > $O/smi_idle.txt 2>&1
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/smi_conv.txt 2>&1 &
timeout 300 python tools/conv_bench.py --voxels 1280000 --reps 400 --only L0_conv3_96_96 > $O/conv.txt 2>&1
wait
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/smi_ubench.txt 2>&1 &
UB_ITERS=200000 timeout 120 tools/bin/coissue > $O/coissue.txt 2>&1
wait
tail -n 3 $O/conv.txt; head -n 30 $O/smi_idle.txt; sed -n 10,30p $O/smi_conv.txt; sed -n 5,25p $O/smi_ubench.txt
timeout 1500 python -m pytest tests/test_gpu_backward.py -x -q > $O/pytest.log 2>&1; tail -n 3 $O/pytest.log
