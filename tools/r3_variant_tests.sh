#!/bin/bash
# GPU box: svgf parity tests with the default library, then bench A/B default vs another build
# usage: tools/r3_variant_tests.sh <out> <other lib>
set -e
R=$(pwd); OUT=$R/gpurun_out/$1; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_svgf_gpu.py tests/test_pipeline_gpu.py tests/test_sharding_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
tools/r3_libs.sh $1 default $2 default $2 default $2
