#!/bin/bash
# NOTE: RMD_ATROUS_SPLIT belonged to a split-column work order that was measured and not committed; kept as
# the recipe of profiles/r03_split_column_ab.txt.
# GPU box: svgf parity tests with the product library, then the frame loop at several sizes and one rank's strip of the 8K frame
# with the split-column a-trous order off and on (experiments build reads RMD_ATROUS_SPLIT), alternating in one call
R=$(pwd); OUT=$R/gpurun_out/${1:-r3split}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_svgf_gpu.py tests/test_pipeline_gpu.py tests/test_sharding_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
export RMD_LIB_PATH=$R/build/variants/librmd_experiments.so
{
for size in "1920 1080" "1280 720" "2560 1440" "3840 2160" "7680 4320"; do
  set -- $size
  for rep in 1 2; do for sp in 0 1; do
    echo -n "split=$sp  "; RMD_ATROUS_SPLIT=$sp PROBE_W=$1 PROBE_H=$2 PROBE_FRAMES=$((1920*1080*120/($1*$2)+8)) timeout -k 10 200 python3 tools/frame_probe.py 2>/dev/null | grep "ms/frame"
  done; done
done
for rep in 1 2; do for sp in 0 1; do
  echo "split=$sp"; RMD_ATROUS_SPLIT=$sp PROBE_EXCHANGE="3" timeout -k 10 300 python3 tools/strip_probe.py 8 4 2>/dev/null | grep -E "one GPU|rank 4"
done; done
} | tee $OUT/ab.txt
