#!/bin/bash
# round 3, GPU session C: k_conv_wl with the deeper pipeline + the one-pass scene-to-click kernel (k_s2c_out): parity + speed
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3c
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -x -q > $OUT/pytest.txt 2>&1; tail -n 4 $OUT/pytest.txt
for CFG in "A3D_NO_WL=1 A3D_FUSED_S2C=0" "A3D_NO_WL=0 A3D_FUSED_S2C=1"; do
  echo "== $CFG" >> $OUT/layers4.txt
  env $CFG LT_BATCH=4 python tools/layer_table.py 2>&1 | grep -v amdgpu >> $OUT/layers4.txt
  echo "== $CFG" >> $OUT/layers1.txt
  env $CFG LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -v amdgpu >> $OUT/layers1.txt
  echo "== $CFG" >> $OUT/bench_quick.txt
  env $CFG python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline 2>&1 | grep '^{' >> $OUT/bench_quick.txt
done
grep -E "sum|==" $OUT/layers4.txt $OUT/layers1.txt
python - <<PY
import json
for l in open("$OUT/bench_quick.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d.get("latency_ms_per_scene"), d.get("decoder_pass_ms_single"), d.get("eval_round_ms"), d["roofline"]["frac"])
    else: print(l.strip())
PY
