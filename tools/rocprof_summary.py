#!/usr/bin/env python3
"""Turn a rocprofv3 output directory (sqlite .db, rocpd format) into the text summaries committed
under profiles/: per-kernel calls / total / average duration (kernel-trace) and per-kernel mean PMC
counter values."""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict


def kernel_stats(db):
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "top_kernels" not in tabs:
        return None
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    return rows


def counters(db):
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tabs:
        return None
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    out = defaultdict(lambda: defaultdict(list))
    for row in cur.execute("select * from counters_collection"):
        d = dict(zip(cols, row))
        out[d.get("kernel_name") or d.get("name")][d.get("counter_name")].append(float(d.get("value") or 0))
    return out


def main():
    res = {}
    for root in sys.argv[1:]:
        for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
            db = sqlite3.connect(path)
            ks = kernel_stats(db)
            if ks:
                print(f"# kernel-trace stats: {path}")
                print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel")
                for name, calls, total, avg, pct in ks:
                    print(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name[:110]}")
            cs = counters(db)
            if cs:
                print(f"# PMC counters (mean per dispatch): {path}")
                for k, c in cs.items():
                    if k and "a3d" in k:
                        short = k.split("(")[0]
                        vals = {n: sum(v) / len(v) for n, v in c.items()}
                        res.setdefault(short, {}).update(vals)
                        print(short, " ".join(f"{n}={v:.5g}(n={len(c[n])})" for n, v in sorted(vals.items())))
    if res and os.environ.get("PMC_JSON"):
        json.dump(res, open(os.environ["PMC_JSON"], "w"), indent=1)


if __name__ == "__main__":
    main()
