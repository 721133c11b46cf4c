#!/usr/bin/env python3
"""pmc_raw.json (per-kernel mean counters from tools/rocprof_summary.py) -> profiles/pmc_summary.json with the
names bench.py uses and the gfx950 corrections of MI355X_MICROARCH.md (HBM section).
    pmc_to_summary.py <pmc_raw.json> <out.json> <workload key> [kernel-trace stats of the same command]
Several template instances fold into one of bench.py's names (k_conv_sk<96,32>: the plain build and the one with the fused
residual projection): with the stats text their counters are averaged by launch count -- per launch of the kernel as
bench.py's `achieved` is --, without it the heaviest instance stands for the name."""
import json
import re
import sys

import os
raw = json.load(open(sys.argv[1]))
wkey = sys.argv[3] if len(sys.argv) > 3 else "80k_b16_q20"       # bench.workload_key of the traced command
full = json.load(open(sys.argv[2])) if os.path.exists(sys.argv[2]) else {}
if "workloads" not in full:
    full = {"workloads": {}}
full["_note"] = ("rocprofv3 --pmc passes of `python bench.py --steps-only --no-profile --steps 5 --warmup 2 --reps 1 --streams 1 "
                 "[--voxels .. --clicks-per-object .. --batch ..]` on MI355X (tools/profile_round.sh), one table per workload "
                 "(<k voxels>k_b<scenes per step>_q<queries per scene>).  Separate passes for FETCH_SIZE, WRITE_SIZE and the SQ "
                 "group (never combined with tracing). hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 "
                 "FETCH_SIZE reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE "
                 "is taken as is (uncalibrated). mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 "
                 "XCDs); bench.py reads hbm_bytes_per_launch of its dominant kernel as roofline.traffic -- only from the table "
                 "of the workload it is running.")
calls = {}
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
    for line in open(sys.argv[4]):
        f = line.split(None, 4)
        if len(f) == 5 and f[0].isdigit():
            calls[f[4].strip().replace("void ", "").replace("a3d::", "").split("(")[0]] = int(f[0])
out = {}
groups = {}
for name, c in raw.items():
    short = name.replace("void ", "").replace("a3d::", "")
    m = re.match(r"k_conv_sk<(\d+), (\d+), (\d+)(?:, (?:true|false))?(?:, \d+)?(?:, (true|false))?>", short)   # <BN, CH, PAIR, DBG>: bench.py's key is <BN,CH>
    key = short
    if m:
        key = f"k_conv_sk<{m.group(1)},{m.group(2)}>" + ("+head" if m.group(4) == "true" else "")
    m = re.match(r"k_dense<(\d+), (\d+)(?:, (?:true|false))?>", short)
    if m:
        key = f"k_dense<{m.group(1)},{m.group(2)}>"
    groups.setdefault(key, []).append((short, c))
    e = out.setdefault(key, {"fetch_kib_raw": 0.0, "write_kib_raw": 0.0, "_n": 0})
    # several template instances can fold into one key: keep the heaviest (largest FETCH_SIZE) as representative
    if c.get("FETCH_SIZE", 0.0) >= e["fetch_kib_raw"]:
        e["fetch_kib_raw"] = c.get("FETCH_SIZE", 0.0)
        e["write_kib_raw"] = c.get("WRITE_SIZE", 0.0)
        e["hbm_bytes_per_launch"] = (2.0 * e["fetch_kib_raw"] + e["write_kib_raw"]) * 1024.0
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        e["mfma_util"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui) if gui else 0.0
        e["lds_bank_conflict_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0)
        e["instance"] = short
    e["_n"] += 1
for key, inst in groups.items():   # launch-weighted mean over the instances of a name
    w = [calls.get(n.split("(")[0], 0) for n, _ in inst]
    if len(inst) > 1 and all(w):
        tot = float(sum(w))
        mean = lambda f: sum(wi * c.get(f, 0.0) for wi, (_, c) in zip(w, inst)) / tot
        e = out[key]
        e["fetch_kib_raw"], e["write_kib_raw"] = mean("FETCH_SIZE"), mean("WRITE_SIZE")
        e["hbm_bytes_per_launch"] = (2.0 * e["fetch_kib_raw"] + e["write_kib_raw"]) * 1024.0
        gui = mean("GRBM_GUI_ACTIVE") / 8.0
        e["mfma_util"] = mean("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * gui) if gui else 0.0
        e["lds_bank_conflict_cycles"] = mean("SQ_LDS_BANK_CONFLICT")
        e["instance"] = " + ".join(f"{wi} x {n.split('(')[0]}" for wi, (n, _) in zip(w, inst))
for e in out.values():
    if isinstance(e, dict):
        e.pop("_n", None)
full["workloads"][wkey] = out
json.dump(full, open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2], wkey, len(out), "kernels")
