#!/bin/bash
# round 3, GPU session A: quick tests of the host-side fixes, workgroup timeline of the dominant conv kernel, share-count
# sweep, real inter-kernel gaps of a step (rocprofv3 timestamps)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3a
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_model.py::test_training_mode_forward_invalidates_the_folded_backbone tests/test_gpu_distributed.py::test_bench_multi_rank_code_path_on_one_gpu tests/test_gpu_conv.py -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
for D in 128 384; do
  A3D_DBG=$D python tools/conv_bench.py --voxels 320000 --only L0_conv3_96_96 --reps 1 2>&1 | grep '^TL' > $OUT/tl_$D.txt
  echo "== A3D_DBG=$D" >> $OUT/timeline.txt
  python tools/wg_timeline.py $OUT/tl_$D.txt >> $OUT/timeline.txt 2>&1
done
cat $OUT/timeline.txt
for G in 0 768 1280 1536 2048 3072 4096; do
  echo "== A3D_SK_G=$G" >> $OUT/sweep_g.txt
  A3D_SK_G=$G python tools/conv_bench.py --voxels 320000 --reps 15 --only conv3_96_96 2>&1 | grep -v amdgpu | grep conv3 >> $OUT/sweep_g.txt
  A3D_SK_G=$G python tools/conv_bench.py --voxels 320000 --reps 15 --only L2_conv3_128_128 2>&1 | grep -v amdgpu | grep conv3 >> $OUT/sweep_g.txt
done
cat $OUT/sweep_g.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/r3a/t4 -o t -- python $R/bench.py --steps-only --streams 1 --steps 8 --reps 1 --warmup 2 --no-profile > $OUT/b4.json 2> $OUT/b4.err
python $R/tools/rocprof_timeline.py /tmp/r3a/t4 k_make_keys 6 > $OUT/timeline_step_4scenes.txt 2>&1
rocprofv3 --kernel-trace -d /tmp/r3a/t1 -o t -- python $R/bench.py --steps-only --streams 1 --batch 1 --steps 8 --reps 1 --warmup 2 --no-profile > $OUT/b1.json 2> $OUT/b1.err
python $R/tools/rocprof_timeline.py /tmp/r3a/t1 k_make_keys 6 > $OUT/timeline_step_1scene.txt 2>&1
tail -3 $OUT/timeline_step_4scenes.txt $OUT/timeline_step_1scene.txt
