#!/bin/bash
# round 3, GPU session D: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG), stage width of the 96-column conv kernels
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3d
mkdir -p $OUT
cd $R
for KA in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$KA" >> $OUT/bench_quick.txt
  HIP_FORCE_DEV_KERNARG=$KA python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline 2>&1 | grep '^{' >> $OUT/bench_quick.txt
done
python - <<PY
import json
for l in open("$OUT/bench_quick.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d.get("latency_ms_per_scene"), d.get("decoder_pass_ms_single"), d.get("eval_round_ms"), d["roofline"]["frac"], d.get("phases_ms_per_step"))
    else: print(l.strip())
PY
for CH in 32 64 96; do
  echo "== A3D_SK_CH=$CH" >> $OUT/conv_ch.txt
  A3D_SK_CH=$CH python tools/conv_bench.py --voxels 320000 --reps 15 --only conv3_96_96 2>&1 | grep -v amdgpu | grep conv3 >> $OUT/conv_ch.txt
  A3D_SK_CH=$CH python tools/conv_bench.py --voxels 320000 --reps 15 --only L0_conv3_128_96 2>&1 | grep -v amdgpu | grep conv3 >> $OUT/conv_ch.txt
done
cat $OUT/conv_ch.txt
cd /tmp && export TMPDIR=/tmp
HIP_FORCE_DEV_KERNARG=1 rocprofv3 --kernel-trace -d /tmp/r3d/t1 -o t -- python $R/bench.py --steps-only --streams 1 --batch 1 --steps 8 --reps 1 --warmup 2 --no-profile > $OUT/b1.json 2> $OUT/b1.err
python $R/tools/rocprof_timeline.py /tmp/r3d/t1 k_make_keys 6 > $OUT/timeline_step_1scene_devkernarg.txt 2>&1
tail -n 2 $OUT/timeline_step_1scene_devkernarg.txt
