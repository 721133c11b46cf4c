#!/bin/bash
# s29: hand-off reduction with 16/NCT parts in flight (full-register builds): 1-scene and 16-scene layer tables, A/B vs tools/bin/lib_base.so
mkdir -p gpurun_out/s29
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -x -q > gpurun_out/s29/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s29/pytest.log
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export A3D_LIB_PATH=$PWD/tools/bin/lib_base.so; else unset A3D_LIB_PATH; fi
  LT_BATCH=1 python tools/layer_table.py > gpurun_out/s29/lt1_${lib}_$rep.txt 2>&1
  LT_BATCH=16 python tools/layer_table.py > gpurun_out/s29/lt16_${lib}_$rep.txt 2>&1
  python bench.py --no-train --steps-only 2>/dev/null | tail -1 > gpurun_out/s29/bench_${lib}_$rep.json
done; done
