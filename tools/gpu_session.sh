#!/bin/bash
# Scratch script of the current GPU session (overwritten per session; `gpurun -- 'bash tools/gpu_session.sh'`).
# Every command under its own `timeout`: a host-side hang otherwise runs into the session limit (s33, r04_experiments.txt).
mkdir -p gpurun_out/check
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/check/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/check/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/check/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/check/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/check/bench.json 2> gpurun_out/check/bench.err
