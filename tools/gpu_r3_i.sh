#!/bin/bash
# round 3, GPU session I: flash-formulation training attention: unit tests vs float64 autograd, the decoder / whole-network gradient tests
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3i
mkdir -p $OUT
cd $R
python -m pytest tests/test_gpu_backward.py -x -q -k "flash" -s > $OUT/pytest_flash.txt 2>&1; tail -n 30 $OUT/pytest_flash.txt
python -m pytest tests/test_gpu_backward.py -x -q -k "decoder_training_step or whole_network or full_training_step or checkpoint or reference_training" -s > $OUT/pytest_dec.txt 2>&1; tail -n 15 $OUT/pytest_dec.txt
python tools/backward_bench.py --step > $OUT/train_step.txt 2>&1; tail -n 12 $OUT/train_step.txt
A3D_TRAIN_FLASH=0 python tools/backward_bench.py --step > $OUT/train_step_noflash.txt 2>&1; tail -n 8 $OUT/train_step_noflash.txt
