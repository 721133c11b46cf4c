# kernel-trace summary of four training iterations (4 x 80k voxels): bash tools/trace_train.sh <tag>
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trt
A3D_TRAIN_TIMING=1 rocprofv3 --kernel-trace --stats -d /tmp/trt -o t -- python $R/tools/backward_bench.py --step --reps 1 > $R/gpurun_out/${TAG}_train_trace.log 2>&1
python $R/tools/rocprof_summary.py /tmp/trt > $R/gpurun_out/${TAG}_training_kernel_trace.txt 2>&1
grep -E "training iteration|train_one_step" $R/gpurun_out/${TAG}_train_trace.log | tail -8
awk 'NR>2{c+=$1; t+=$2} END{print "launches", c, "total_us", t}' $R/gpurun_out/${TAG}_training_kernel_trace.txt
head -40 $R/gpurun_out/${TAG}_training_kernel_trace.txt | cut -c1-150
