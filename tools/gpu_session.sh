#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_try.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_try.json").read().strip().splitlines()[-1])
print(round(d["value"],1), d["latency_ms_per_scene"], d["value_batch4"], d["train_iter"])
PY
