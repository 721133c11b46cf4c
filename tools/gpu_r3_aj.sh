#!/bin/bash
# round 3, GPU session AJ: is the texture-address / vector-cache path (TA / TCP) what the 96-column conv kernel waits on?
# The gathered A fragments are "fragment-shaped" loads (16 rows x 64 B per instruction: every 128-B line is touched twice),
# the weight DMA adds 8 lines per piece.  Counters in their own passes (no trace domains besides --kernel-trace).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/aj
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TA|TCP|TD|TCC)_[A-Z0-9_a-z]+" | sort -u > $OUT/avail_counters.txt
PMC="python $R/bench.py --steps-only --no-profile --steps 3 --warmup 1 --reps 1 --streams 1"
run() {   # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/aj/$name -o p -- $PMC > /dev/null 2> $OUT/$name.err
  python $R/tools/rocprof_summary.py /tmp/aj/$name 2>&1 | grep -E "k_conv_sk<96|k_conv_sk<128, 32, 1|k_kv_c2s|k_s2c_out|k_dense" > $OUT/$name.txt
  cat $OUT/$name.txt
}
run ta1 TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run tcp2 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum
run tcp3 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum
run sq1 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL
ls $OUT
