#!/bin/bash
# the current GPU session's command list (overwritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
timeout 300 tools/bin/coissue > $O/coissue.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --voxels 300000 --clicks-per-object 4 --batch 1 --streams 2 --steps 10 --warmup 3 --reps 7 --no-train > $O/bench_config5.json 2> $O/bench5.err
python - <<'P'
import json
for f in ('bench','bench_config5'):
    d=json.loads(open(f'gpurun_out/s10/{f}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print(f, d['value'], d.get('value_batch4'), r['frac'], r['traffic'], r.get('hbm_traffic_frac'), r.get('profiles_workload'), r.get('rocprof_avg_launch_us'), r.get('agrees_with_profiles_within_10pct'))
P
tail -n 20 $O/coissue.txt
