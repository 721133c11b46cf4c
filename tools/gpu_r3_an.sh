#!/bin/bash
# round 3, GPU session AN: hand-off without the agent-scope acquire (parts read with L1-bypassing agent-scope loads) and with more
# parts in flight on the high-register builds.  Same-box A/B against the previous build (tools/bin/libbase.so).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/an
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_backward.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
for lib in base new base new; do
  [ $lib = base ] && export A3D_LIB_PATH=$R/tools/bin/libbase.so || unset A3D_LIB_PATH
  echo "== $lib: one scene"; LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -E "L4  256-> 256|L3  128-> 128|L2   64->  64|L3  256-> 256|L2  128-> 128|sum" | awk '{k=$4" "$5" "$6" "$7" "$8; if ($1=="sum") print; else {s[k]+=$(NF-3); n[k]++}} END {for (k in s) printf "   %s  avg %.1f us over %d\n", k, s[k]/n[k], n[k]}'
done
for lib in base new; do
  [ $lib = base ] && export A3D_LIB_PATH=$R/tools/bin/libbase.so || unset A3D_LIB_PATH
  echo "== $lib: 16 scenes"; LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "sum"
  echo "== $lib: 4 scenes"; LT_BATCH=4 python tools/layer_table.py 2>&1 | grep -E "sum"
done
for lib in base new base new; do
  [ $lib = base ] && export A3D_LIB_PATH=$R/tools/bin/libbase.so || unset A3D_LIB_PATH
  echo "== bench $lib"
  python bench.py --no-cpu-baseline --reps 7 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('latency_ms_per_scene'), d.get('decoder_pass_ms_single'), d.get('eval_round_ms'))"
done
