#!/usr/bin/env python3
"""Per-PHASE kernel summary of traced training iterations.  train_one_step launches one marker kernel (torch.cuda._sleep ->
`spin_kernel`) at every phase mark when A3D_TRAIN_TIMING=mark, with the device synchronised on both sides, so the dispatches
between two markers are exactly one phase.  Reads a rocprofv3 --kernel-trace database.
   python tools/train_phase_trace.py <rocprofv3 output dir> [iterations to skip = 1] [rows per phase = 14]
A3D_PHASE_GAPS=<phase index>: also list the longest idle stretches of that phase in the first iteration counted."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

PHASES = ["backbone forward", "click simulation", "decoder forward", "losses", "decoder backward", "backbone backward",
          "clip + AdamW"]


def short(n):
    n = n.replace("a3d::", "").replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0] if not n.startswith("at::") else n[:70]
    return n[:60]


def main():
    root = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(path).cursor()
        views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        for v in ([x for x in views if x == "kernels"] or [x for x in views if "kernel" in x.lower()]):
            cols = [r[1] for r in cur.execute(f"pragma table_info({v})")]
            if not {"start", "end", "name"} <= set(cols):
                continue
            rows = list(cur.execute(f"select name, start, end from {v} order by start"))
            if not rows:
                continue
            n_mark = sum(1 for r in rows if "spin_kernel" in r[0])
            per_iter = len(PHASES) + 1
            print(f"# {path}: {len(rows)} dispatches, {n_mark} markers = {n_mark / per_iter:.2f} iterations; first {skip} skipped")
            agg = [defaultdict(lambda: [0, 0.0]) for _ in PHASES]
            wall = [0.0] * len(PHASES)
            union = [0.0] * len(PHASES)      # time with at least one kernel running (streams overlap: <= kernel sum, <= span)
            busy = [0.0] * len(PHASES)
            launches = [0] * len(PHASES)
            m = -1                      # markers seen - 1: the segment behind marker m is phase m % per_iter of iteration m // per_iter
            seg = []
            done_iters = set()

            def close():
                if m < 0 or not seg:
                    return
                p, it = m % per_iter, m // per_iter
                if p >= len(PHASES) or it < skip:
                    return
                done_iters.add(it)
                if os.environ.get("A3D_PHASE_GAPS") == str(p) and it == skip:      # the idle stretches of one phase of one iteration
                    gaps, end_, prev = [], seg[0][2], seg[0][0]
                    for name, s_, e_ in seg[1:]:
                        if s_ > end_:
                            gaps.append(((s_ - end_) / 1e3, (s_ - seg[0][1]) / 1e3, short(prev), short(name)))
                        if e_ > end_:
                            end_, prev = e_, name
                    print(f"# idle stretches of phase '{PHASES[p]}', iteration {it}: {sum(g[0] for g in gaps):.1f} us in {len(gaps)} gaps; the longest:")
                    for g in sorted(gaps, reverse=True)[:25]:
                        print(f"#   {g[0]:8.1f} us at {g[1]:9.1f} us   {g[2]} -> {g[3]}")
                wall[p] += (max(r[2] for r in seg) - seg[0][1]) / 1e3
                end = seg[0][1]
                for _, s_, e_ in seg:              # sorted by start
                    if e_ > end:
                        union[p] += (e_ - max(s_, end)) / 1e3
                        end = e_
                for name, s_, e_ in seg:
                    a_ = agg[p][short(name)]
                    a_[0] += 1
                    a_[1] += (e_ - s_) / 1e3
                    busy[p] += (e_ - s_) / 1e3
                    launches[p] += 1
            for r in rows:
                if "spin_kernel" in r[0]:
                    close()
                    m += 1
                    seg = []
                else:
                    seg.append(r)
            iters = max(len(done_iters), 1)
            print(f"# averaged over {iters} iterations")
            tot_l = tot_b = tot_w = 0
            for p, nm in enumerate(PHASES):
                print(f"== {nm}: {launches[p] / iters:.0f} launches, kernel sum {busy[p] / iters / 1e3:.2f} ms, "
                      f"device busy {union[p] / iters / 1e3:.2f} ms, first-start..last-end {wall[p] / iters / 1e3:.2f} ms per iteration")
                tot_l += launches[p] / iters
                tot_b += busy[p] / iters / 1e3
                tot_w += wall[p] / iters / 1e3
                for k, (c, t) in sorted(agg[p].items(), key=lambda kv: -kv[1][1])[:top]:
                    print(f"   {c / iters:8.1f} x {t / c:9.1f} us = {t / iters / 1e3:7.3f} ms  {k}")
            print(f"== total: {tot_l:.0f} launches, kernel sum {tot_b:.2f} ms, phase spans {tot_w:.2f} ms per iteration")
            break


if __name__ == "__main__":
    main()
