#!/bin/bash
# round 3, GPU session V: packed query-side weights + batched click rounds: tests, training iterations, bench
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3v
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_clicks.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -n 12 > $OUT/pytest.txt
cat $OUT/pytest.txt
A3D_DEC_DBG=2 LT_BATCH=1 python tools/layer_table.py 2>&1 | grep "k_query_layer dbg" | tail -n 2
LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "query_chain|sum" | tail -n 5
A3D_TRAIN_TIMING=1 A3D_BB_ITERS=10 timeout 900 python tools/backward_bench.py > $OUT/train.txt 2>&1
grep -E "training iteration|train_one_step" $OUT/train.txt | tail -n 20
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("latency_ms_per_scene"), d.get("decoder_pass_ms_single"), d.get("eval_round_ms"), d["roofline"]["frac"], d.get("eval_rounds_per_s"), d.get("iou_at_k",{}).get("match"))
PY
