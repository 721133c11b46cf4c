#!/bin/bash
# the current GPU session's command list (overwritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_distributed.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for cfg in "--scenes 4 --voxels 5000 --lr 1e-3 --iters 400 --every 100" "--scenes 4 --voxels 5000 --lr 3e-4 --iters 400 --every 100" "--scenes 4 --voxels 5000 --lr 1e-3 --iters 400 --every 100 --colour 0.5" "--scenes 2 --voxels 5000 --lr 1e-3 --iters 300 --every 100 --batch 1"; do
  echo "== $cfg" >> $O/fit.log
  timeout 600 python tools/fit_synthetic.py $cfg 2>&1 | grep -v amdgpu.ids >> $O/fit.log
done
cat $O/fit.log
