#!/bin/bash
# round 3, GPU session P: the bench line of the final build (default command)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3p
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "
import sys,json
d=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','latency_ms_per_scene','decoder_pass_ms_single','eval_round_ms','eval_rounds_per_s','pipeline_frac','pipeline_hbm_frac','max_abs_diff')})
print(d['roofline']); print(d.get('iou_at_k'))"
python bench.py --gpus 2 --steps 5 --warmup 2 --reps 3 --no-profile > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; tail -c 600 $OUT/bench_2ranks.json
