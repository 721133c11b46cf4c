#!/usr/bin/env python3
"""Per-workgroup timeline of k_conv_sk (DBG build, A3D_DBG=128 or 384):
    A3D_DBG=384 python tools/conv_bench.py --voxels 320000 --only L0_conv3_96_96 --reps 1 2>&1 | grep '^TL' > tl.txt
    python tools/wg_timeline.py tl.txt [launches_to_skip]
Every launch prints one line per workgroup; the LAST launch in the file is analysed: kernel span, when workgroups end
(tail), per-CU occupancy over time, share of a workgroup's life spent at stage ends (vmcnt + barrier) and in hand-off
waits."""
import collections
import sys

import numpy as np

rows = [l.split() for l in open(sys.argv[1]) if l.startswith("TL")]
R = np.array([[int(r[i]) for i in (2, 4, 6, 8, 10, 12, 14, 16, 18)] for r in rows], dtype=np.int64)
# split into launches: ticket numbers restart
w = R[:, 0]
G = int(w.max()) + 1
n_launch = len(R) // G
R = R[-G:] if len(R) % G == 0 else R[-G:]
w, xcc, hw, t0, t1, stages, tiles, wait, hand = R.T
clk = 100e6   # s_memtime ticks: constant 100 MHz on gfx9 (REFCLK); printed both ways below
span = t1.max() - t0.min()
print(f"launches in file {n_launch}, workgroups {G}; span {span} ticks")
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
ncu = len(set(cu.tolist()))
life = t1 - t0
print(f"CUs seen {ncu}; workgroup life: min {life.min()} mean {life.mean():.0f} max {life.max()} ticks "
      f"({life.mean() / span:.3f} of the span on average)")
print(f"start skew: last start {t0.max() - t0.min()} ticks after the first; "
      f"end times (fraction of span): p10 {np.percentile(t1 - t0.min(), 10) / span:.3f} p50 "
      f"{np.percentile(t1 - t0.min(), 50) / span:.3f} p90 {np.percentile(t1 - t0.min(), 90) / span:.3f} "
      f"p99 {np.percentile(t1 - t0.min(), 99) / span:.3f}")
print(f"stages per workgroup: min {stages.min()} mean {stages.mean():.1f} max {stages.max()}; tiles mean {tiles.mean():.1f}")
print(f"ticks per stage (life / stages): mean {(life / np.maximum(stages, 1)).mean():.1f}")
if wait.sum():
    print(f"stage-end wait (vmcnt + barrier, wave 0): {wait.sum() / life.sum():.3f} of workgroup life")
print(f"hand-off wait: {hand.sum() / life.sum():.3f} of workgroup life; max {hand.max()} ticks")
# resident workgroups per CU over time -> average number of running workgroups during the span
ev = sorted([(a, 1) for a in t0] + [(b, -1) for b in t1])
cur, last, area = 0, ev[0][0], 0
hist = collections.Counter()
for t, d in ev:
    area += cur * (t - last)
    hist[cur * 4 // max(G, 1)] += t - last
    last, cur = t, cur + d
print(f"average workgroups alive during the span: {area / span:.1f} of {G} ({area / span / G:.3f})")
per_cu_end = np.array([t1[cu == c].max() for c in set(cu.tolist())]) - t0.min()
print(f"per-CU last end / span: min {per_cu_end.min() / span:.3f} mean {per_cu_end.mean() / span:.3f}")
per_cu_stages = np.array([stages[cu == c].sum() for c in set(cu.tolist())])
print(f"stages per CU: min {per_cu_stages.min()} mean {per_cu_stages.mean():.1f} max {per_cu_stages.max()} "
      f"(max/mean {per_cu_stages.max() / per_cu_stages.mean():.3f})")
per_cu_wgs = collections.Counter(cu.tolist())
print("workgroups per CU:", sorted(collections.Counter(per_cu_wgs.values()).items()))
