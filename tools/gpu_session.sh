#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/d15
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 $O/gpu_tests.log
