#!/bin/bash
# round 3, GPU session AC: whole GPU suite + smoke + default bench on the current build
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3ac
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 12 > $OUT/pytest.txt
cat $OUT/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -n 2 $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("latency_ms_per_scene"), d.get("decoder_pass_ms_single"), d.get("eval_round_ms"), d["roofline"]["frac"], d.get("eval_rounds_per_s"), d["iou_at_k"]["rounds_with_identical_iou"], d["iou_at_k"]["rounds"])
PY
