#!/usr/bin/env python3
"""kernel-trace stats text (tools/rocprof_summary.py) -> profiles/kernel_avg_us.json: average launch duration in us per
kernel under the names bench.py uses (k_conv_sk<BN,CH>, k_dense<NS,NCT>, others by bare name), stored under the workload
key given as the third argument (bench.workload_key: "<k voxels>k_b<batch>_q<queries>"; other workloads already in the
output file are kept).  bench.py compares its live HIP-event average of the dominant kernel with the table of the
workload it runs (roofline.agrees_with_profiles_within_10pct).
    kernel_avg.py <stats.txt> <out.json> <workload key>"""
import json
import re
import sys

import os
wkey = sys.argv[3] if len(sys.argv) > 3 else "80k_b16_q20"
full = json.load(open(sys.argv[2])) if os.path.exists(sys.argv[2]) else {}
if "workloads" not in full:
    full = {"workloads": {}}
full["_note"] = ("rocprofv3 --kernel-trace --stats average launch duration (us) of `bench.py --steps-only --streams 1` per workload "
                 "(<k voxels>k_b<scenes per step>_q<queries per scene>); made by tools/kernel_avg.py")
out = {"_source": sys.argv[1].split("/")[-1]}
acc = {}
for line in open(sys.argv[1]):
    f = line.split(None, 4)
    if len(f) < 5 or not f[0].isdigit():
        continue
    calls, total = int(f[0]), float(f[1])
    name = f[4].strip().replace("void ", "").replace("a3d::", "")
    m = re.match(r"k_conv_sk<(\d+), (\d+), (\d+)(?:, (?:true|false))?(?:, \d+)?(?:, (true|false))?>", name)
    mw = re.match(r"k_conv_wl<(\d+), (\d+)(?:, \d+)?>", name)
    if m:
        key = f"k_conv_sk<{m.group(1)},{m.group(2)}>" + ("+head" if m.group(4) == "true" else "")   # <BN, CH, PAIR, FUSE, STATS, HEAD>
    elif mw:
        key = f"k_conv_wl<{mw.group(1)},{mw.group(2)}>"
    else:
        m = re.match(r"k_dense<(\d+), (\d+)(?:, (?:true|false))?>", name)
        key = f"k_dense<{m.group(1)},{m.group(2)}>" if m else name.split("(")[0]
    c, t = acc.get(key, (0, 0.0))
    acc[key] = (c + calls, t + total)
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    out[k] = round(t / c, 3)
full["workloads"][wkey] = out
json.dump(full, open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2], len(acc), "kernels")
