#!/bin/bash
# round 3, GPU session F: conv kernel with the tile's gather rows staged in LDS: parity + speed; SQ counters of the big kernels
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3f
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_scene.py -x -q > $OUT/pytest.txt 2>&1; tail -n 4 $OUT/pytest.txt
LT_BATCH=4 python tools/layer_table.py 2>&1 | grep -v amdgpu > $OUT/layers4.txt
LT_BATCH=1 python tools/layer_table.py 2>&1 | grep -v amdgpu > $OUT/layers1.txt
cat $OUT/layers4.txt | head -66 | tail -64; tail -n 1 $OUT/layers4.txt $OUT/layers1.txt
python bench.py --steps 20 --warmup 5 --reps 9 --no-cpu-baseline 2>&1 | grep '^{' > $OUT/bench_quick.txt
python - <<PY
import json
for l in open("$OUT/bench_quick.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d.get("latency_ms_per_scene"), d.get("decoder_pass_ms_single"), d.get("eval_round_ms"), d["roofline"]["frac"], d.get("phases_ms_per_step"))
PY
cd /tmp && export TMPDIR=/tmp
PMC="python $R/bench.py --steps-only --no-profile --steps 4 --warmup 2 --reps 1 --streams 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/r3f/pmc_sq -o p -- $PMC > /dev/null 2> $OUT/pmc_sq.err
PMC_JSON=$OUT/pmc_sq_raw.json python $R/tools/rocprof_summary.py /tmp/r3f/pmc_sq > $OUT/pmc_sq.txt 2>&1
head -40 $OUT/pmc_sq.txt
