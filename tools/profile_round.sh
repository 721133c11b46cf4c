#!/bin/bash
# Regenerate the profiles/ artefacts of a round on the GPU box:  bash tools/profile_round.sh r02
# (run through gpurun; raw rocprofv3 databases stay in /tmp, text/JSON summaries go to gpurun_out/<tag>/).
# Passes: kernel trace + stats of the DEFAULT bench command (4 scenes x 4 streams in flight + the instrumented pass) and
# of `bench.py --steps-only --streams 1` (only the 4-scene steps, stream-serial: kernels never overlap and every launch
# of a kernel is a launch of the step the bench line's roofline object describes -> kernel_avg_us.json), then SEPARATE
# --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ group) -- counters are never combined with the hip/hsa/memory trace domains.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/$TAG/trace -o t -- python $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
python $R/tools/rocprof_summary.py /tmp/$TAG/trace > $OUT/kernel_trace_stats.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/$TAG/trace1 -o t -- python $R/bench.py --steps-only --streams 1 > $OUT/bench_under_rocprof_steps_only.json 2> $OUT/trace1.err
python $R/tools/rocprof_summary.py /tmp/$TAG/trace1 > $OUT/kernel_trace_stats_steps_only.txt 2>&1
cp $R/profiles/kernel_avg_us.json $OUT/kernel_avg_us.json 2>/dev/null; cp $R/profiles/pmc_summary.json $OUT/pmc_summary.json 2>/dev/null
python $R/tools/kernel_avg.py $OUT/kernel_trace_stats_steps_only.txt $OUT/kernel_avg_us.json 80k_b16_q20
PMC="python $R/bench.py --steps-only --no-profile --steps 5 --warmup 2 --reps 1 --streams 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/$TAG/pmc_fetch -o p -- $PMC > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/$TAG/pmc_write -o p -- $PMC > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --kernel-trace -d /tmp/$TAG/pmc_sq -o p -- $PMC > /dev/null 2> $OUT/pmc_sq.err
PMC_JSON=$OUT/pmc_raw.json python $R/tools/rocprof_summary.py /tmp/$TAG/pmc_fetch /tmp/$TAG/pmc_write /tmp/$TAG/pmc_sq > $OUT/pmc_counters.txt 2>&1
python $R/tools/pmc_to_summary.py $OUT/pmc_raw.json $OUT/pmc_summary.json 80k_b16_q20 $OUT/kernel_trace_stats_steps_only.txt
# the same passes on BASELINE.json config 5 (one 300 k-voxel scene, 20 clicks): its own tables in the two JSON files
C5="--voxels 300000 --clicks-per-object 4 --batch 1"
rocprofv3 --kernel-trace --stats -d /tmp/$TAG/trace5 -o t -- python $R/bench.py --steps-only --streams 1 $C5 > $OUT/bench_under_rocprof_steps_only_config5.json 2> $OUT/trace5.err
python $R/tools/rocprof_summary.py /tmp/$TAG/trace5 > $OUT/kernel_trace_stats_steps_only_config5.txt 2>&1
python $R/tools/kernel_avg.py $OUT/kernel_trace_stats_steps_only_config5.txt $OUT/kernel_avg_us.json 300k_b1_q30
PMC5="python $R/bench.py --steps-only --no-profile --steps 10 --warmup 2 --reps 1 --streams 1 $C5"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/$TAG/pmc5_fetch -o p -- $PMC5 > /dev/null 2> $OUT/pmc5_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/$TAG/pmc5_write -o p -- $PMC5 > /dev/null 2> $OUT/pmc5_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --kernel-trace -d /tmp/$TAG/pmc5_sq -o p -- $PMC5 > /dev/null 2> $OUT/pmc5_sq.err
PMC_JSON=$OUT/pmc_raw_config5.json python $R/tools/rocprof_summary.py /tmp/$TAG/pmc5_fetch /tmp/$TAG/pmc5_write /tmp/$TAG/pmc5_sq > $OUT/pmc_counters_config5.txt 2>&1
python $R/tools/pmc_to_summary.py $OUT/pmc_raw_config5.json $OUT/pmc_summary.json 300k_b1_q30 $OUT/kernel_trace_stats_steps_only_config5.txt
cp $OUT/kernel_avg_us.json $OUT/pmc_summary.json $R/profiles/      # so that the config-5 line below reads ITS tables
# BASELINE.json config 5 (KITTI-like 300 k voxels, 20 clicks): its own bench line and per-launch table
# (four steps in flight like the headline; 40 steps per repetition: with 10 the fill and drain of the four streams are a tenth of the region)
python $R/bench.py --voxels 300000 --clicks-per-object 4 --batch 1 --streams 4 --steps 40 --warmup 4 --reps 7 --no-train > $OUT/bench_config5.json 2> $OUT/bench_config5.err
LT_BATCH=1 LT_VOXELS=300000 LT_CPO=4 python $R/tools/layer_table.py > $OUT/layer_table_config5.txt 2>&1
LT_BATCH=4 python $R/tools/layer_table.py > $OUT/layer_table_4scenes.txt 2>&1
LT_BATCH=1 python $R/tools/layer_table.py > $OUT/layer_table_1scene.txt 2>&1
LT_BATCH=16 python $R/tools/layer_table.py > $OUT/layer_table_16scenes.txt 2>&1
# decoder pass by query count (5 objects x LT_CPO clicks + 10 learned queries; one 80 k scene)
for CPO in 2 5 10 11 15 22 30 38; do
  echo "== clicks per object $CPO" >> $OUT/decoder_by_queries.txt
  LT_CPO=$CPO LT_BATCH=1 python $R/tools/layer_table.py 2>&1 | awk '/posenc/{p=1} p' >> $OUT/decoder_by_queries.txt
done
# PMC counters of the decoder's wide-tier kernels (85 and 160 queries)
bash $R/tools/pmc_wide.sh $TAG/pmc_wide_q85 15 > /dev/null 2>&1; cp $OUT/pmc_wide_q85/pmc_wide.txt $OUT/pmc_wide_q85.txt 2>/dev/null
bash $R/tools/pmc_wide.sh $TAG/pmc_wide_q160 30 > /dev/null 2>&1; cp $OUT/pmc_wide_q160/pmc_wide.txt $OUT/pmc_wide_q160.txt 2>/dev/null
cd /tmp
# the lock-step evaluation round on a fitted state dict next to the random-init one (20 / 60 / 100 queries)
python $R/tools/eval_rounds_fitted.py 800 > $OUT/eval_rounds_fitted.txt 2> $OUT/eval_rounds_fitted.err
# eight data-parallel ranks on the one GPU (gloo): the real training iteration with overlapped buckets + SyncBN
(cd $R && python -m pytest tests/test_gpu_distributed.py -q -k eight_rank_dp -s 2>&1 | tail -5) > $OUT/dp8_training_one_gpu.txt
# the fork analysis over several fit lengths
python $R/tools/fork_hunt.py 40 60 80 100 140 180 > $OUT/fork_evidence.txt 2> $OUT/fork_evidence.err
# training iterations (4 x 80 k voxels, the real train_one_step): phase times, then plain wall clock
A3D_BB_ITERS=10 A3D_TRAIN_TIMING=1 python $R/tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration|train_one_step" > $OUT/training_iterations.txt
A3D_BB_ITERS=10 python $R/tools/backward_bench.py --step --reps 1 2>&1 | grep -E "training iteration" >> $OUT/training_iterations.txt
# kernel trace of four training iterations
rm -rf /tmp/$TAG/trt
A3D_TRAIN_TIMING=1 rocprofv3 --kernel-trace --stats -d /tmp/$TAG/trt -o t -- python $R/tools/backward_bench.py --step --reps 1 > $OUT/train_trace.log 2>&1
python $R/tools/rocprof_summary.py /tmp/$TAG/trt > $OUT/training_kernel_trace.txt 2>&1
python $R/tools/rocprof_stalls.py /tmp/$TAG/trt 10 > $OUT/training_stalls.txt 2>&1
# the same four iterations (+ two warm-ups) split by PHASE: one marker dispatch per phase boundary (A3D_TRAIN_TIMING=mark)
rm -rf /tmp/$TAG/trp
A3D_BB_ITERS=6 A3D_TRAIN_TIMING=mark rocprofv3 --kernel-trace -d /tmp/$TAG/trp -o t -- python $R/tools/backward_bench.py --step --reps 1 > $OUT/train_phase_trace.log 2>&1
python $R/tools/train_phase_trace.py /tmp/$TAG/trp 2 40 > $OUT/training_phases.txt 2>&1
# eight ranks sharing the one GPU (host-side check of an 8-rank run) next to one rank driving the same 16 scenes per step
A3D_BENCH_ONE_GPU=1 python $R/bench.py --gpus 8 --batch 2 --steps 20 --warmup 3 --reps 3 --no-profile --steps-only --no-train --no-cpu-baseline > $OUT/bench_8ranks_one_gpu.json 2> $OUT/bench_8ranks.err
python $R/bench.py --gpus 1 --batch 16 --steps 20 --warmup 3 --reps 3 --no-profile --steps-only --no-train --no-cpu-baseline > $OUT/bench_1rank_same_load.json 2> $OUT/bench_1rank.err
# config 5 at round 3's protocol too (two steps in flight, 10 steps per repetition)
python $R/bench.py --voxels 300000 --clicks-per-object 4 --batch 1 --streams 2 --steps 10 --warmup 4 --reps 7 --no-train --no-profile --steps-only > $OUT/bench_config5_r3_protocol.json 2> $OUT/bench_config5_r3.err
# steps per repetition: what the fill and drain of the four streams cost at the driver's 20 steps
for K in 20 80; do echo -n "steps=$K: "; python $R/bench.py --steps-only --no-profile --steps $K --reps 9 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],1), 'scenes/s', round(r['ms_per_step'],3), 'ms per step')"; done > $OUT/steps_per_repetition.txt 2>&1
ls -la $OUT
