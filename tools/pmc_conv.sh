# SQ counters of one conv layer (default L0 3^3 96->96) -- separate --pmc passes, kernel trace only.
#   CASE=L1_conv3_96_96 VOX=320000 bash tools/pmc_conv.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$n -o p -- python $R/tools/conv_bench.py --voxels ${VOX:-80000} --only ${CASE:-L0_conv3_96_96} --reps 5 > /tmp/pmc_$n.log 2>&1; python $R/tools/pmc_summary.py /tmp/pmc_$n 2>&1 | grep -E "spconv|k_conv|DURATION" ; }
run a SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY
run e GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU
run f SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
