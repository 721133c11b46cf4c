#!/bin/bash
# s28: full GPU suite + bench on the build with the decoder-tape changes and the coarse click stage
mkdir -p gpurun_out/s28
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s28/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s28/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s28/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/s28/smoke.log
python bench.py > gpurun_out/s28/bench.json 2> gpurun_out/s28/bench.err
A3D_BB_ITERS=14 python tools/backward_bench.py --step --reps 1 2>&1 | grep "training iteration" | cut -c1-70 > gpurun_out/s28/train.log
