#!/bin/bash
# round 3, GPU session H: static wave priority in the conv kernel (A3D_SK_PRIO), k_s2c_out with the mask embeddings requested
# before the Y stores; then the batch x streams sweep
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3h
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_model.py -x -q -k "golden or one_pass or oracle" > $OUT/pytest.txt 2>&1; tail -n 3 $OUT/pytest.txt
for P in 0 1; do
  echo "== A3D_SK_PRIO=$P" >> $OUT/conv_prio.txt
  A3D_SK_PRIO=$P python tools/conv_bench.py --voxels 320000 --reps 15 --only conv3_96_96 2>&1 | grep -v amdgpu | grep conv3 >> $OUT/conv_prio.txt
  A3D_SK_PRIO=$P python tools/conv_bench.py --voxels 320000 --reps 15 --only L0_conv3_128_96 2>&1 | grep -v amdgpu | grep conv3 >> $OUT/conv_prio.txt
  A3D_SK_PRIO=$P python tools/conv_bench.py --voxels 320000 --reps 15 --only L2_conv3_128_128 2>&1 | grep -v amdgpu | grep conv3 >> $OUT/conv_prio.txt
  echo "== A3D_SK_PRIO=$P" >> $OUT/bench_quick.txt
  A3D_SK_PRIO=$P python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline 2>&1 | grep '^{' >> $OUT/bench_quick.txt
done
cat $OUT/conv_prio.txt
python - <<PY
import json
for l in open("$OUT/bench_quick.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d.get("latency_ms_per_scene"), d.get("decoder_pass_ms_single"), d.get("eval_round_ms"), d["roofline"]["frac"], d["kernels_ms_per_step"].get("s2c_attn"), d["kernels_ms_per_step"].get("c2s_attn"))
    else: print(l.strip())
PY
bash tools/gpu_r3_g.sh
