#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/d16
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 $O/gpu_tests.log
python tools/latency_one_scene.py --deep 1
python tools/latency_one_scene.py --deep 1
timeout 600 python bench.py --no-cpu-baseline --no-train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', round(d['value'],1), 'lat', d['latency_ms_per_scene'], 'dec_single', d.get('decoder_pass_ms_single'), 'eval_round', d.get('eval_round_ms'), 'eval_rounds_per_s', d.get('eval_rounds_per_s'), 'b4', d.get('value_batch4'))"
