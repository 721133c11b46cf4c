#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python tools/qc_probe.py 2>&1 | grep -v amdgpu | tail -45
