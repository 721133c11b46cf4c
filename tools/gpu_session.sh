#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/d23
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fit.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?"
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
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trp
A3D_BB_ITERS=6 A3D_TRAIN_TIMING=mark rocprofv3 --kernel-trace -d /tmp/trp -o t -- python $R/tools/backward_bench.py --step --reps 1 > $R/$O/train_phase_trace.log 2>&1
python $R/tools/train_phase_trace.py /tmp/trp 2 60 > $R/$O/training_phases.txt 2>&1
grep "^==" $R/$O/training_phases.txt
