#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/s19; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
cd /tmp; export TMPDIR=/tmp
for lib in prev new prev new; do
  if [ $lib = prev ]; then export A3D_LIB_PATH=$R/agile3d_amd/libagile3d_hip_prev.so; else unset A3D_LIB_PATH; fi
  timeout 900 python $R/bench.py --no-cpu-baseline > $O/bench_$lib.json 2> $O/bench_$lib.err
  python - <<PY
import json
d = json.loads([l for l in open("$O/bench_$lib.json") if l.startswith("{")][-1])
f = d.get("iou_at_k", {})
print("$lib", d["value"], d["latency_ms_per_scene"], d["decoder_pass_ms_single"], d["eval_round_ms"], d["eval_rounds_per_s"], d["value_batch4"],
      d["train_iter"]["ms_without_click_rounds"], d["train_iter"]["ms_per_click_round"], f.get("rounds_with_identical_clicks"), (f.get("forks") or {}).get("unexplained"))
PY
done
