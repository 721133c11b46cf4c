#!/bin/bash
# s24: host threads capped to the container's CPU quota: training iterations and the bench line, new vs old behaviour
mkdir -p gpurun_out/s24
{
for v in "X=0" "A3D_HOST_THREADS=0" "X=0"; do
  echo "== $v"; env $v bash tools/cpu_quota_probe.sh env A3D_BB_ITERS=14 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|cpu.max|usage_usec" | cut -c1-110
done
} > gpurun_out/s24/train.log 2>&1
for v in "X=0" "A3D_HOST_THREADS=0"; do
  echo "== $v"; env $v bash tools/cpu_quota_probe.sh python bench.py 2>&1 | grep -E "^\{|usage_usec"
done > gpurun_out/s24/bench.log 2>&1
