#!/bin/bash
# GPU box: quick A/B -- svgf parity tests, then bench serial and two-stream pipelined (no CPU baseline / other sizes)
set -e
R=$(pwd); OUT=$R/gpurun_out/${1:-r3ab}; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_svgf_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for mode in 0 1; do
  RMD_PIPELINE=$mode timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-sizes > $OUT/bench_p$mode.json 2> $OUT/bench_p$mode.err || { tail -5 $OUT/bench_p$mode.err; exit 1; }
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench_p$mode.json'))
print('pipeline=$mode', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['per_iteration_ms'])"
done
