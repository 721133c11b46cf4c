#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (sqlite .db or csv) per kernel: mean counter value per dispatch."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    out = defaultdict(lambda: defaultdict(list))
    if "counters_collection" in tabs:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        q = "select * from counters_collection"
        for row in cur.execute(q):
            d = dict(zip(cols, row))
            name = d.get("kernel_name") or d.get("name")
            out[name][d.get("counter_name")].append(float(d.get("value") or d.get("counter_value") or 0))
    return out, tabs


def main():
    for root in sys.argv[1:]:
        for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
            out, tabs = from_db(path)
            if "top_kernels" in tabs:   # kernel-trace durations next to the counters (clock = GRBM_GUI_ACTIVE / duration)
                db = sqlite3.connect(path)
                for name, calls, total, avg, pct in db.execute(
                        "select name, total_calls, total_duration, average, percentage from top_kernels"):
                    if "a3d" in name:
                        print(f"DURATION {name.split('(')[0][-40:]} calls={calls} avg_ns={avg:.1f}")
            if not out:
                print(path, "no counters_collection; tables:", tabs[:40])
            for k, cs in out.items():
                if k and ("a3d" in k):
                    short = k.split("(")[0][-40:]
                    print(short, " ".join(f"{c}={sum(v) / len(v):.4g}(n={len(v)})" for c, v in sorted(cs.items())))


if __name__ == "__main__":
    main()
