#!/bin/bash
# s33: HIP runtime launch-path settings on the one-scene latency and the batched step (same box A/B)
mkdir -p gpurun_out/s33
run() { echo -n "$1: "; env $1 python bench.py --no-train --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), 'scenes/s | latency', r.get('latency_ms_per_scene'), '| decoder pass', r.get('decoder_pass_ms_single'), '| eval round', r.get('eval_round_ms'), '| batch4', r.get('value_batch4'))"; }
{ run X=0; run HIP_FORCE_DEV_KERNARG=1; run HIP_FORCE_DEV_KERNARG=0; run ROC_SYSTEM_SCOPE_SIGNAL=0; run X=0; run HIP_FORCE_DEV_KERNARG=1; run HIP_FORCE_DEV_KERNARG=0; run DEBUG_HIP_KERNARG_COPY_OPT=0; } > gpurun_out/s33/launch_path.log 2>&1
