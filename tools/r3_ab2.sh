#!/bin/bash
# GPU box: full gpu tests then bench (4K + other sizes, no CPU baseline)
set -e
R=$(pwd); OUT=$R/gpurun_out/${1:-r3d}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('$OUT/bench.json'))
r=d['roofline']
print('4K %.1f Mpix/s  %.4f ms (median %s)  A %s' % (d['value'], d['ms_per_step'], d['ms_per_step_median'], ' '.join('%.1f' % (1e3*v) for v in r['per_iteration_ms'])))
print({k: (v['mpix_s'], v['ms_per_frame']) for k, v in d['other_sizes'].items()}, d['cornell_sequence_4k']['fps'])"
