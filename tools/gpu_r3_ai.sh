#!/bin/bash
# round 3, GPU session AI: the decoder / click / training tests with NaN-poisoned allocations (reads of memory nothing wrote)
R=$GRAFT_REPO_ROOT
cd $R
A3D_POISON=1 timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_clicks.py tests/test_gpu_backward.py -m gpu -q 2>&1 | tail -n 15
