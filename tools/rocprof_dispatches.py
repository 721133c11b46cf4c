#!/usr/bin/env python3
"""Per-dispatch durations (us) of the kernels whose name contains a substring, in launch order:
   python tools/rocprof_dispatches.py <rocprofv3 output dir> <substring> [first] [count]"""
import glob, os, sqlite3, sys
root, sub = sys.argv[1], sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 60
for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    db = sqlite3.connect(path)
    cur = db.cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cand = [v for v in views if v == "kernels"] or [v for v in views if "kernel" in v.lower()]
    for v in cand:
        cols = [r[1] for r in cur.execute(f"pragma table_info({v})")]
        if "start" in cols and "end" in cols and "name" in cols:
            rows = list(cur.execute(f"select name, start, end from {v} where name like ? order by start", (f"%{sub}%",)))
            print(f"# {path} {v}: {len(rows)} dispatches of *{sub}*")
            for name, s, e in rows[first:first + count]:
                print(f"{(e - s) / 1e3:9.2f} us  gap_before {0:6.1f}  {name[:60]}")
            break
    else:
        print("no kernel view with start/end in", path, views[:20])
