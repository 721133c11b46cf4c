#!/bin/bash
# round 3, GPU session Z: multi-block query layer on the second build (parity, decoder pass by query count, training iterations)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3z
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -n 8
for CPO in 15 30; do
  echo "== clicks per object $CPO" >> $OUT/dec_by_queries.txt
  LT_CPO=$CPO LT_BATCH=1 python tools/layer_table.py 2>&1 | awk '/posenc/{p=1} p' | grep -E "query_chain|sum" >> $OUT/dec_by_queries.txt
done
cat $OUT/dec_by_queries.txt
A3D_BB_ITERS=10 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" > $OUT/train_c.txt
cat $OUT/train_c.txt | cut -c1-120
