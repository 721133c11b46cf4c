#!/bin/bash
# Scratch script of the current GPU session (overwritten per session; `gpurun -- 'bash tools/gpu_session.sh'`).
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/final
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 3000 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -3 $O/gpu_tests.log
timeout 5400 bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
ls gpurun_out/r05 | wc -l
