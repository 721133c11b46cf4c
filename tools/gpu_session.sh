#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/d20
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 $O/gpu_tests.log
python tools/latency_one_scene.py --deep 1
A3D_SORT_ONE_LAUNCH=0 python tools/latency_one_scene.py --deep 1
python tools/latency_one_scene.py --deep 1
A3D_SORT_ONE_LAUNCH=0 python tools/latency_one_scene.py --deep 1
for m in 1 0 1 0; do
A3D_SORT_ONE_LAUNCH=$m timeout 300 python bench.py --batch 4 --steps-only --no-profile --reps 7 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch4 x4 streams one_launch=$m', round(d['value'],1))"
done
for m in 1 0; do
A3D_SORT_ONE_LAUNCH=$m timeout 300 python bench.py --batch 1 --steps-only --no-profile --reps 7 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch1 x4 streams one_launch=$m', round(d['value'],1))"
done
