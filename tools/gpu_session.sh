#!/bin/bash
# the current GPU session's command list (overwritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --voxels 300000 --clicks-per-object 4 --batch 1 --streams 2 --steps 10 --warmup 3 --reps 7 --no-train --no-cpu-baseline > $O/bench_config5.json 2> $O/bench5.err
LT_BATCH=16 timeout 300 python tools/layer_table.py > $O/layer_table_16.txt 2>&1
LT_BATCH=1 LT_VOXELS=300000 LT_CPO=4 timeout 300 python tools/layer_table.py > $O/layer_table_config5.txt 2>&1
python - <<'P'
import json
for f in ('bench','bench_config5'):
    d=json.loads(open(f'gpurun_out/s5/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d.get('value_batch4'), d.get('latency_ms_per_scene'), d.get('kernels_ms_per_step'), d.get('conv_tflops'))
P
grep -E "k_dense|s2c_attn|c2s_attn|ln_mask" $O/layer_table_16.txt $O/layer_table_config5.txt
