#!/usr/bin/env python3
"""One step's launches in stream order from a rocprofv3 --kernel-trace database: start offset, duration, gap to the previous
kernel's end.   python tools/rocprof_timeline.py <rocprofv3 output dir> [anchor substring] [which occurrence] [count]
The window starts at the given occurrence of a kernel whose name contains the anchor (default k_make_keys = first kernel
of a scene build)."""
import glob, os, sqlite3, sys
root = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_make_keys"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 5
count = int(sys.argv[4]) if len(sys.argv) > 4 else 140
for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    db = sqlite3.connect(path)
    cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    for v in ([x for x in views if x == "kernels"] or [x for x in views if "kernel" in x.lower()]):
        cols = [r[1] for r in cur.execute(f"pragma table_info({v})")]
        if not {"start", "end", "name"} <= set(cols):
            continue
        rows = list(cur.execute(f"select name, start, end from {v} order by start"))
        idx = [i for i, r in enumerate(rows) if anchor in r[0]]
        if len(idx) <= which:
            print("anchor not found often enough", len(idx))
            break
        i0 = idx[which]
        i1 = idx[which + 1] if len(idx) > which + 1 else min(len(rows), i0 + count)
        base = rows[i0][1]
        prev_end = None
        tot_k = tot_gap = 0.0
        for name, s, e in rows[i0:i1]:
            gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
            short = name.replace("void a3d::", "").replace("a3d::", "").replace("(anonymous namespace)::", "")[:70]
            print(f"{(s - base) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.2f}  gap {gap:7.2f}  {short}")
            tot_k += (e - s) / 1e3
            tot_gap += max(gap, 0.0)
            prev_end = e if prev_end is None else max(prev_end, e)
        print(f"# window: {i1 - i0} launches, kernels {tot_k:.1f} us, gaps {tot_gap:.1f} us, wall {(rows[i1 - 1][2] - base) / 1e3:.1f} us")
        break
