#!/usr/bin/env python3
"""Analyse the per-tile timeline k_spconv2 dumps with A3D_DBG=4096:
    A3D_DBG=4096 python tools/conv_bench.py --only L0_conv3_96_96 --reps 1 2>&1 | grep '^TT' > tt.txt
    python tools/tile_timeline.py tt.txt 1246
Prints prologue / stage loop / epilogue cycles per tile, how many tiles a CU has in flight, per-CU spans."""
import collections
import sys

import numpy as np

rows = [l.split() for l in open(sys.argv[1])]
if len(sys.argv) > 2:
    rows = rows[-int(sys.argv[2]):]
T = np.array([[int(r[i]) for i in (1, 3, 5, 7, 9, 11, 12, 13, 14)] for r in rows], dtype=np.int64)
tile, wg, hwid, xcc, st, t0, t1, t2, t3 = T.T
cu = ((hwid >> 8) & 0xf) | (((hwid >> 12) & 1) << 4) | (((hwid >> 13) & 7) << 5) | (xcc << 8)
print("per tile (cycles): prologue %.0f, loop %.0f (%.0f per stage, %.1f stages), epilogue %.0f" % (
    (t1 - t0).mean(), (t2 - t1).mean(), ((t2 - t1) / np.maximum(st, 1)).mean(), st.mean(), (t3 - t2).mean()))
spans, inflight = [], []
for c in set(cu.tolist()):
    m = cu == c
    spans.append((t3[m].max() - t0[m].min()) / 2400.0)
    ev = sorted([(a, 1) for a in t0[m]] + [(b, -1) for b in t3[m]])
    cur, last, tim = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        tim[min(cur, 3)] += t - last
        last, cur = t, cur + d
    tot = sum(tim.values())
    inflight.append([tim[k] / tot for k in range(4)])
spans = np.array(spans)
print("CUs %d, workgroups %d; per-CU span us min %.1f mean %.1f max %.1f" % (len(spans), len(set(wg.tolist())), spans.min(),
                                                                           spans.mean(), spans.max()))
print("fraction of a CU's span with 0/1/2/3+ tiles in flight:", np.array(inflight).mean(0).round(3))
