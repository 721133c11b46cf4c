#!/bin/bash
# s30: steps in flight for the small-batch protocols (batch 4 x 80 k, config 5 = batch 1 x 300 k)
mkdir -p gpurun_out/s30
for s in 4 6 8 4 6 8; do
  echo -n "batch4 streams=$s: "; python bench.py --steps-only --no-train --batch 4 --streams $s --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), r['ms_per_step'])"
  echo -n "config5 streams=$s: "; python bench.py --steps-only --no-train --voxels 300000 --clicks-per-object 4 --batch 1 --streams $s --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), r['ms_per_step'])"
done > gpurun_out/s30/streams.log 2>&1
