#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/d26
mkdir -p $O
timeout 3000 python -m pytest tests/test_gpu_model.py tests/test_gpu_clicks.py tests/test_gpu_fit.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/tests.log
python tools/qc_probe.py 2>&1 | grep -v amdgpu | sed -n 1,12p
CPO=8 python tools/qc_probe.py 2>&1 | grep -v amdgpu | sed -n 1,15p
CPO=30 python tools/qc_probe.py 2>&1 | grep -v amdgpu | grep -E "query_chain|back-to-back"
