#!/bin/bash
# round 3, GPU session T: second build of the single-block query layer (k_query_block) -- parity, phase marks, latency
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3t
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_clicks.py -m gpu -x -q 2>&1 | tail -n 12 > $OUT/pytest.txt
cat $OUT/pytest.txt
A3D_DEC_DBG=2 LT_BATCH=1 python tools/layer_table.py 2>&1 | grep "k_query_layer dbg" | tail -n 3 > $OUT/ql_dbg.txt
cat $OUT/ql_dbg.txt
LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "query_chain|sum" | tail -n 5
A3D_QL_V1=1 LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "query_chain|sum" | tail -n 5
