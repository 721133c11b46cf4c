#!/bin/bash
# round 3, GPU session R: decoder kernels with global (not flat) loads; 128-column conv kernels on 32-channel stages
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3r
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -n 8 > $OUT/pytest.txt
cat $OUT/pytest.txt
LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "posenc|query_chain|c2s_attn|s2c_attn|sum" > $OUT/layers1.txt
LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "spconv<128>|c2s_attn|s2c_attn|query_chain|sum" > $OUT/layers16.txt
tail -n 14 $OUT/layers1.txt
grep -E "c2s_attn|s2c_attn|query_chain|sum" $OUT/layers16.txt | tail -n 8
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("latency_ms_per_scene"), d.get("decoder_pass_ms_single"), d.get("eval_round_ms"), d["roofline"]["frac"], d.get("eval_rounds_per_s"), d.get("iou_at_k",{}).get("match"))
PY
