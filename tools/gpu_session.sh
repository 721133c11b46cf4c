#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/d6
mkdir -p $O
python tools/latency_one_scene.py --deep 0
python tools/latency_one_scene.py --deep 1
python tools/latency_one_scene.py --deep 0
python tools/latency_one_scene.py --deep 1
timeout 900 python tools/conv_bench.py --sweep --reps 10 --only conv3_ > $O/sweep80.txt 2>&1
grep -v "^L0\|^L1" $O/sweep80.txt
timeout 900 python tools/conv_bench.py --sweep --reps 10 --voxels 300000 --only "L2_\|L3_\|L4_" > $O/sweep300.txt 2>&1
