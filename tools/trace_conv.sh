# kernel-trace durations (no event overhead) of conv_bench cases:  VOX=320000 CASES="L4_conv3 L3_conv3" bash tools/trace_conv.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in ${CASES:-L4_conv3_256_256 L3_conv3_256_256 L2_conv3_128_128 L2_conv3_64_64}; do
  rm -rf /tmp/trc
  rocprofv3 --kernel-trace --stats -d /tmp/trc -o t -- python $R/tools/conv_bench.py --voxels ${VOX:-320000} --only $C --reps 10 > /tmp/trc.log 2>&1
  echo "== $C A3D_CONV_SK=${A3D_CONV_SK:-1} A3D_SK_MINSHARE=${A3D_SK_MINSHARE:-} A3D_SK_OV=${A3D_SK_OV:-}"
  python $R/tools/rocprof_summary.py /tmp/trc 2>&1 | grep -E "k_conv_sk|k_spconv2|splitk" | cut -c1-100
done
