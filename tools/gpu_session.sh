#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/d21
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_clicks.py tests/test_gpu_backward.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/tests.log
for CPO in 15 30; do
  echo "== clicks per object $CPO"
  LT_CPO=$CPO LT_BATCH=1 python tools/layer_table.py 2>&1 | awk '/posenc/{p=1} p' | cut -c1-75 | sed -n 1,12p
  LT_CPO=$CPO LT_BATCH=1 python tools/layer_table.py 2>&1 | tail -1
done
python - <<'PY'
import sys, torch
sys.argv=["bench.py"]
import bench
r = bench.train_iter_ms(torch.device("cuda"))
print(r["ms_without_click_rounds"], r["ms_per_click_round"], r["ms_all"], r["phases_ms_median"])
PY
