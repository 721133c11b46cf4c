for D in 0 1 2 3 4 7 64 65; do echo "PAIR=0 DBG=$D: $(A3D_SK_PAIR=0 A3D_DBG=$D python tools/conv_bench.py --voxels 320000 --reps 10 --only L0_conv3_96_96 2>&1 | grep -v amdgpu | tail -1 | cut -c1-75)"; done
echo "PAIR=1 (product): $(python tools/conv_bench.py --voxels 320000 --reps 10 --only L0_conv3_96_96 2>&1 | grep -v amdgpu | tail -1 | cut -c1-75)"
