#!/bin/bash
# the current GPU session's command list (overwritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_fit.py -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/s3/bench.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('iou_at_k'))[:3000]); print(d['value'], d.get('value_batch4'), d.get('roofline',{}).get('traffic'), d.get('roofline',{}).get('profiles_workload'))
P
