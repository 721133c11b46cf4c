#!/bin/bash
# the current GPU session's command list (overwritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s16; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
timeout 3000 bash tools/profile_round.sh r04c > $O/profile_round.log 2>&1
tail -n 3 $O/profile_round.log
