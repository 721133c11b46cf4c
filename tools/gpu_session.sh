#!/bin/bash
# the current GPU session's command list (overwritten per session; results land in gpurun_out/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
timeout 300 tools/bin/coissue > $O/coissue.txt 2>&1
cat $O/coissue.txt
