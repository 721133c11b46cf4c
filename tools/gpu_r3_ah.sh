#!/bin/bash
# round 3, GPU session AH: batched position encodings
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_clicks.py -m gpu -x -q 2>&1 | tail -n 4
LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "posenc|sum" | tail -n 4
python bench.py --steps 20 --warmup 5 --reps 7 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('latency_ms_per_scene'), d.get('decoder_pass_ms_single'), d.get('eval_round_ms'), d.get('eval_rounds_per_s'))"
