#!/bin/bash
# Scratch script of the current GPU session (overwritten per session; `gpurun -- 'bash tools/gpu_session.sh'`).
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 $O/gpu_tests.log
A3D_POISON=1 timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_conv.py tests/test_gpu_scene.py -q -x > $O/poison_tests.log 2>&1; echo "poison tests rc=$?"
tail -3 $O/poison_tests.log
