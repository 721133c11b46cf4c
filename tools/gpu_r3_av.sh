#!/bin/bash
# round 3, GPU session AV: the scene-to-click query projection of the unfused decoder half (> 64 queries) on a second stream
# (A3D_DEC_INNER=0: after the query-side chain, on the group's own stream)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/av
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_clicks.py tests/test_gpu_backward.py -m gpu -x -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
A3D_POISON=1 timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "many_clicks or batched_decoder or scene_cache" 2>&1 | tail -2
python - <<'PY'
import os, time, torch, numpy as np, subprocess, sys
PY
for v in 0 1 0 1; do
  echo "== A3D_DEC_INNER=$v"
  A3D_DEC_INNER=$v python - <<'PY'
import time, torch, numpy as np
from agile3d_amd import SparseTensor, build_model, default_args, randomize_bn_stats
from agile3d_amd.synthetic import make_scene, make_clicks
torch.manual_seed(0)
model = randomize_bn_stats(build_model(default_args())).eval().cuda()
sc = make_scene(80000, seed=0)
x = SparseTensor(features=torch.from_numpy(sc["feats"]), coordinates=torch.from_numpy(sc["coords"]), device="cuda")
raw = torch.from_numpy(sc["raw_xyz"]).cuda()
r = model.forward_backbone(x, raw_coordinates=raw)
for per in (15, 30):
    ci, ct = make_clicks(sc["labels"], 5, per, 0, seed=1)
    for _ in range(4): model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct])
    torch.cuda.synchronize(); ts = []
    for _ in range(30):
        t0 = time.perf_counter(); model.forward_mask(*r, click_idx=[ci], click_time_idx=[ct]); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("queries", 5 * per + 10, "decoder pass %.3f ms" % float(np.median(ts)))
PY
  A3D_DEC_INNER=$v A3D_BB_ITERS=10 python tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" | awk '{print $3, $4}' | tr '\n' ' '; echo
done
