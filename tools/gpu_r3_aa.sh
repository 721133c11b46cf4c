#!/bin/bash
# round 3, GPU session AA: fused click-to-scene kernel at 48 / 64 queries
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3aa
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -n 8
for CPO in 5 10; do
  echo "== clicks per object $CPO" >> $OUT/dec_by_queries.txt
  LT_CPO=$CPO LT_BATCH=1 python tools/layer_table.py 2>&1 | awk '/posenc/{p=1} p' >> $OUT/dec_by_queries.txt
done
cat $OUT/dec_by_queries.txt
