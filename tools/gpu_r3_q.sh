#!/bin/bash
# round 3, GPU session Q: 128-column conv kernels on the 16-scene step: stage width / register build
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3q
mkdir -p $OUT
cd $R
for CFG in "A3D_SK_CH128=0" "A3D_SK_CH128=32" "A3D_SK_CH128=32 A3D_SK_PAIR=1" "A3D_SK_PAIR=1"; do
  echo "== $CFG" >> $OUT/layers16.txt
  env $CFG LT_BATCH=16 python tools/layer_table.py 2>&1 | grep -E "spconv<128>|spconv< 64>|sum" >> $OUT/layers16.txt
done
python - <<PY
import re
blocks={}; cur=None
for l in open("$OUT/layers16.txt"):
    if l.startswith("=="): cur=l.strip(); blocks[cur]=[]; continue
    blocks[cur].append(l.rstrip())
names=list(blocks)
base=blocks[names[0]]
for i,l in enumerate(base):
    row=[l[:74]]
    for n in names[1:]:
        m=re.search(r"([\d.]+) us", blocks[n][i]) if i < len(blocks[n]) else None
        row.append(m.group(1) if m else blocks[n][i][-20:] if i < len(blocks[n]) else "")
    print(" | ".join(row))
print(names)
PY
