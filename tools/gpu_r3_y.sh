#!/bin/bash
# round 3, GPU session Y: decoder pass by query count (one 80 k scene; 5 objects x LT_CPO clicks + 10 learned queries)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3y
mkdir -p $OUT
cd $R
for CPO in 2 5 10 15 30; do
  echo "== clicks per object $CPO" >> $OUT/dec_by_queries.txt
  LT_CPO=$CPO LT_BATCH=1 python tools/layer_table.py 2>&1 | awk '/posenc/{p=1} p' >> $OUT/dec_by_queries.txt
done
cat $OUT/dec_by_queries.txt
