#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/d24
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fit.py tests/test_gpu_distributed.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?"
tail -3 $O/tests.log
for i in 1 2; do
python - <<'PY'
import sys, torch
sys.argv=["bench.py"]
import bench
r = bench.train_iter_ms(torch.device("cuda"))
print(r["ms_without_click_rounds"], r["ms_per_click_round"], r["phases_ms_median"])
PY
done
