#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/s15; mkdir -p $O
cd $R
timeout 1700 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "error_bound" > $O/emu_bound.txt 2>&1; echo "emu bound rc=$?"; tail -1 $O/emu_bound.txt
timeout 2400 python -m pytest tests/test_gpu_clicks.py tests/test_gpu_fit.py tests/test_gpu_model.py tests/test_gpu_backward.py -x -q -m gpu -k "eval or valid or Eval or fit or epoch or click" > $O/eval_tests.txt 2>&1; echo "eval tests rc=$?"; tail -2 $O/eval_tests.txt
cd /tmp; export TMPDIR=/tmp
timeout 1200 python $R/bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
for k in ("value", "latency_ms_per_scene", "decoder_pass_ms_single", "eval_round_ms", "eval_rounds_per_s", "value_batch4"):
    print(k, d.get(k))
print(json.dumps(d.get("emulated_fp32_products"), indent=1)[:1500])
print(d["roofline"]["frac"], d["train_iter"]["ms_without_click_rounds"], d["train_iter"]["ms_per_click_round"])
PY
