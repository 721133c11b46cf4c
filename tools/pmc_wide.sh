#!/bin/bash
# PMC passes over the decoder's wide-tier kernels (one 80 k scene, LT_CPO clicks per object):  bash tools/pmc_wide.sh <out> [cpo]
# Counters in their own runs with --kernel-trace only (never combined with the hip / hsa trace domains).
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcw}; CPO=${2:-15}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/layer_table.py"
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf /tmp/pw$i
  LT_CPO=$CPO LT_BATCH=1 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pw$i -o p -- $CMD > /dev/null 2> $OUT/pmc$i.err
done
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/pw1 /tmp/pw2 /tmp/pw3 /tmp/pw4 2>&1 | grep -E "k_c2s_w|k_s2c_w|k_out_w|k_query_block|k_kv_c2s|k_c2s_combine" | grep -v "^ *[0-9]" > $OUT/pmc_wide.txt
cat $OUT/pmc_wide.txt
