#!/bin/bash
# the current GPU session's command list (overwritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 3 $O/pytest.log
LT_BATCH=16 timeout 300 python tools/layer_table.py > $O/layer_table_16.txt 2>&1
LT_BATCH=1 LT_VOXELS=300000 LT_CPO=4 timeout 300 python tools/layer_table.py > $O/layer_table_c5.txt 2>&1
grep -E "s2c_attn|c2s_attn|^sum" $O/layer_table_16.txt $O/layer_table_c5.txt
timeout 600 python bench.py --no-cpu-baseline --no-train > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/s11/bench.json').read().strip().splitlines()[-1])
print(d['value'], d.get('value_batch4'), d.get('latency_ms_per_scene'), d['roofline']['frac'], d.get('kernels_ms_per_step'))
P
