#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
A3D_DEC_DBG=2 LT_CPO=15 LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "k_query_layer dbg" | tail -n 2
A3D_QL_V1=1 LT_CPO=15 LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "query_chain" | tail -n 2
cd /tmp && export TMPDIR=/tmp && LT_CPO=15 LT_BATCH=1 rocprofv3 --kernel-trace --stats -d /tmp/qtr -o t -- python $R/tools/layer_table.py > /dev/null 2>&1
python $R/tools/rocprof_summary.py /tmp/qtr 2>/dev/null | grep -E "k_query|k_c2s_combine|calls" | head
