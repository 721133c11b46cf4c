#!/bin/bash
# round 3, GPU session O: the whole -m gpu suite, smoke, then the profile round (bench line, traces, counters, config 5)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3o
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -n 5 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
bash tools/profile_round.sh r03 > $OUT/profile_round.log 2>&1; tail -n 30 $OUT/profile_round.log
cat $R/gpurun_out/r03/bench.json | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','latency_ms_per_scene','decoder_pass_ms_single','eval_round_ms','eval_rounds_per_s','pipeline_frac','pipeline_hbm_frac','max_abs_diff')})
print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline'].get('value_incl_kernel_maps')); print(d.get('iou_at_k')); print(d.get('emulated_fp32_products'))"
