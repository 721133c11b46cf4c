#!/bin/bash
# round 3, GPU session E: decoder kernels with explicit LDS prefetch of the weight fragments: parity + speed
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3e
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_model.py -x -q > $OUT/pytest.txt 2>&1; tail -n 4 $OUT/pytest.txt
LT_BATCH=4 python tools/layer_table.py 2>&1 | grep -v amdgpu > $OUT/layers4.txt
LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -v amdgpu > $OUT/layers1.txt
tail -n 19 $OUT/layers4.txt; tail -n 14 $OUT/layers1.txt
python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline 2>&1 | grep '^{' > $OUT/bench_quick.txt
python - <<PY
import json
for l in open("$OUT/bench_quick.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d.get("latency_ms_per_scene"), d.get("decoder_pass_ms_single"), d.get("eval_round_ms"), d["roofline"]["frac"], d.get("phases_ms_per_step"))
PY
