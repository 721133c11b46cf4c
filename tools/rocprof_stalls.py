#!/usr/bin/env python3
"""Where a traced run stalls: the longest single dispatches and the longest device-idle gaps (no kernel of any stream running)
of a rocprofv3 --kernel-trace database, each with the kernels on either side.
   python tools/rocprof_stalls.py <rocprofv3 output dir> [how many]"""
import glob, os, sqlite3, sys
root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    db = sqlite3.connect(path)
    cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    for v in ([x for x in views if x == "kernels"] or [x for x in views if "kernel" in x.lower()]):
        cols = [r[1] for r in cur.execute(f"pragma table_info({v})")]
        if not {"start", "end", "name"} <= set(cols):
            continue
        rows = list(cur.execute(f"select name, start, end from {v} order by start"))
        if not rows:
            continue
        t0 = rows[0][1]
        short = lambda n: n.replace("a3d::", "").replace("void ", "").replace("(anonymous namespace)::", "")[:48]
        print(f"# {path}: {len(rows)} dispatches over {(rows[-1][2] - t0) / 1e6:.1f} ms")
        print("longest dispatches (at ms, us, kernel):")
        for name, s, e in sorted(rows, key=lambda r: r[1] - r[2])[:top]:
            print(f"  {(s - t0) / 1e6:10.2f} ms  {(e - s) / 1e3:10.1f} us  {short(name)}")
        gaps = []
        busy_end, last = rows[0][2], rows[0][0]
        for name, s, e in rows[1:]:
            if s > busy_end:
                gaps.append((s - busy_end, busy_end, last, name))
            if e > busy_end:
                busy_end, last = e, name
        tot = sum(g[0] for g in gaps)
        print(f"device idle {tot / 1e6:.1f} ms in {len(gaps)} gaps; longest (at ms, us, after -> before):")
        for g, at, a, b in sorted(gaps, reverse=True)[:top]:
            print(f"  {(at - t0) / 1e6:10.2f} ms  {g / 1e3:10.1f} us  {short(a)} -> {short(b)}")
        break
