#!/bin/bash
# round 3, GPU session AQ: bounded click search, phase A skipped on small samples; sampling stride sweep
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/aq
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_clicks.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for s in 8 16 32 64; do
  A3D_CLICK_SAMPLE=$s timeout 300 python -m pytest tests/test_gpu_clicks.py -m gpu -x -q -k "bounded or kdtree" > $OUT/tests_s$s.log 2>&1
  echo "tests stride $s rc=$?"
  for cfg in "80000 0.05" "80000 0.3" "80000 0.6" "300000 0.1" "300000 0.3" "300000 0.6"; do
    echo "== stride $s: $cfg"; A3D_CLICK_SAMPLE=$s python tools/click_bench.py $cfg 2>&1 | grep median
  done
done
echo "== plain"
for cfg in "80000 0.05" "300000 0.1"; do A3D_CLICK_PRUNE=0 python tools/click_bench.py $cfg 2>&1 | grep median; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/aq/tr -o t -- python $R/tools/click_bench.py 300000 0.3 > /dev/null 2> $OUT/trace.err
python $R/tools/rocprof_summary.py /tmp/aq/tr 2>&1 | head -14
