#!/bin/bash
# PMC passes over the weight-gradient kernels (4 x 80 k voxels, tools/backward_bench.py):  bash tools/pmc_wgrad.sh <out>
# Counters in their own runs with --kernel-trace only (never combined with the hip / hsa trace domains).
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcwg}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/backward_bench.py --batch 4 --reps 3 --only ${2:-0}"
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf /tmp/pg$i
  rocprofv3 --pmc $SET --kernel-trace -d /tmp/pg$i -o p -- $CMD > /dev/null 2> $OUT/pmc$i.err
done
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py /tmp/pg1 /tmp/pg2 /tmp/pg3 /tmp/pg4 2>&1 | grep -E "k_wgrad" | grep -v "^ *[0-9]" > $OUT/pmc_wgrad.txt
cat $OUT/pmc_wgrad.txt
