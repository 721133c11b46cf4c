#!/bin/bash
# round 3, GPU session S: phase marks of the query-side layer kernel (one scene, 20 queries)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3s
mkdir -p $OUT
cd $R
A3D_DEC_DBG=2 LT_BATCH=1 python tools/layer_table.py 2>&1 | grep "k_query_layer dbg" | tail -n 4 > $OUT/ql_dbg.txt
cat $OUT/ql_dbg.txt
A3D_QL_HELPERS=1 A3D_DEC_DBG=2 LT_BATCH=1 python tools/layer_table.py 2>&1 | grep "k_query_layer dbg" | tail -n 2
