#!/bin/bash
# ablations of the conv kernel on one layer (A3D_DBG bits: 1 no A gather, 2 no weight DMA, 4 no MFMA)
OUT=gpurun_out/diag_conv.txt
: > $OUT
for V in ${VOXS:-320000}; do
 for RG in 1 2; do
 for D in 0 1 2 3 4 7; do
  echo "== voxels $V RG=$RG A3D_DBG=$D" >> $OUT
  A3D_SK_RG=$RG A3D_DBG=$D python tools/conv_bench.py --voxels $V --reps 10 --only ${CASE:-L0_conv3_96_96} 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT
 done
 done
done
