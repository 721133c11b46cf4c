#!/bin/bash
# Scratch script of the current GPU session (overwritten per session; `gpurun -- 'bash tools/gpu_session.sh'`).
R=$GRAFT_REPO_ROOT
cd $R
timeout 5400 bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
ls gpurun_out/r05 | wc -l
