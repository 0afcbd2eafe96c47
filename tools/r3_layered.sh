#!/bin/bash
# NOTE: RMD_ATROUS_LAYERED existed in commit 0bc5408 only (the order was measured and taken out again).
# GPU box: svgf parity tests with the product library, then the frame loop at 1080p / 720p / 1440p / 4K with the layered a-trous
# order off and on (experiments build reads RMD_ATROUS_LAYERED), alternating in one call
R=$(pwd); OUT=$R/gpurun_out/${1:-r3lay}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_svgf_gpu.py tests/test_pipeline_gpu.py tests/test_sharding_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
export RMD_LIB_PATH=$R/build/variants/librmd_experiments.so
for size in "1920 1080" "1280 720" "2560 1440" "3840 2160"; do
  set -- $size
  for rep in 1 2; do for lay in 0 1; do
    echo -n "layered=$lay  "; RMD_ATROUS_LAYERED=$lay PROBE_W=$1 PROBE_H=$2 PROBE_FRAMES=120 timeout -k 10 200 python3 tools/frame_probe.py 2>/dev/null | grep "ms/frame"
  done; done
done | tee $OUT/ab.txt
