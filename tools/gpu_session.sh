#!/bin/bash
# s35: full GPU suite on the build with the fused residual projections, then the round's profile set
mkdir -p gpurun_out/s35
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s35/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s35/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s35/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/s35/smoke.log
timeout 2700 bash tools/profile_round.sh r04e > /dev/null 2>&1
