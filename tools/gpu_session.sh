#!/bin/bash
# final validation of the round: smoke, the whole GPU suite, then every r05 artefact regenerated from HEAD
set -u
R=$(pwd); O=$R/gpurun_out/final; mkdir -p $O
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
timeout 2400 bash tools/profile_round.sh r05 > $R/gpurun_out/r05_profile_round.log 2>&1; echo "profile_round rc=$?"
