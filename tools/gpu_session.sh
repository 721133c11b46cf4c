#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s15; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_conv.py tests/test_gpu_scene.py tests/test_gpu_model.py tests/test_gpu_backward.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log
A3D_GRID=0 timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_scene.py -x -q 2>&1 | tail -n 2
LT_BATCH=16 timeout 300 python tools/layer_table.py 2>&1 | grep -E "^ *(0|1|2) |^sum"
LT_BATCH=1 timeout 300 python tools/layer_table.py 2>&1 | grep -E "^ *(0|1|2) |^sum"
timeout 600 python bench.py --steps-only --no-profile --reps 9 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
