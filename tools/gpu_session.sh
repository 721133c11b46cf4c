#!/bin/bash
# the current GPU session's command list (overwritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
for RS in 4 2; do
  A3D_CONV_RS=$RS A3D_CONV_RS_MIN=1 timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -x -q > $O/pytest_rs$RS.log 2>&1; echo "rs$RS pytest rc=$?" | tee -a $O/pytest_rs$RS.log
  tail -3 $O/pytest_rs$RS.log
done
for RS in 0 4 2 0 4 2; do
  echo "== A3D_CONV_RS=$RS" | tee -a $O/conv.log
  A3D_CONV_RS=$RS timeout 600 python tools/conv_bench.py --voxels 1280000 --reps 10 --only conv3_ 2>&1 | grep -E "L0_conv3_96_96|L0_conv3_128_96|L1_conv3_96_96" | tee -a $O/conv.log
done
for RS in 4 2; do
  echo "== one scene A3D_CONV_RS=$RS MIN=1" | tee -a $O/conv.log
  A3D_CONV_RS=$RS A3D_CONV_RS_MIN=1 timeout 600 python tools/conv_bench.py --voxels 80000 --reps 20 --only conv3_ 2>&1 | grep -E "L0_conv3_96_96|L0_conv3_128_96|L1_conv3_96_96" | tee -a $O/conv.log
done
A3D_CONV_RS=0 timeout 600 python tools/conv_bench.py --voxels 80000 --reps 20 --only conv3_ 2>&1 | grep -E "L0_conv3_96_96|L0_conv3_128_96|L1_conv3_96_96" | tee -a $O/conv.log
