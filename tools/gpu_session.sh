#!/bin/bash
set -u
R=$(pwd); O=$R/gpurun_out/s16; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
ER_BATCH=1 ER_ROUNDS=12 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o er -- python $R/tools/eval_round_probe.py > $O/prof.log 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$O/prof/*.db")[0])
c = db.cursor()
rows = list(c.execute("select name, start, end from kernels order by start"))
# the last full round: from the last k_query_init to the end
idx = [i for i, r in enumerate(rows) if "k_query_init" in r[0]]
a = idx[-2]; b = idx[-1]
t0 = rows[a][1]
for r in rows[a:b]:
    print("%8.1f %7.1f  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:90]))
PY
