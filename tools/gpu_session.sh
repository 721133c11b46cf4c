#!/bin/bash
set -u
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for s in 4 6 8 3 4; do
  echo "streams $s: $(timeout 600 python $R/bench.py --steps-only --no-profile --streams $s 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
