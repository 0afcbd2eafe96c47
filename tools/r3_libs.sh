#!/bin/bash
# GPU box: bench.py (4K, no extras) once per library given; prints value / ms / per-iteration a-trous times
# usage: tools/r3_libs.sh <outname> <lib or "default"> ...
R=$(pwd); OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset RMD_LIB_PATH; else export RMD_LIB_PATH=$R/$lib; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-sizes ${BENCH_ARGS} > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -5 $OUT/bench_$name.err; continue; }
  python3 -c "
import json
d=json.load(open('$OUT/bench_$name.json'))
r=d['roofline']
print('%-28s %8.1f Mpix/s  %.4f ms (median %s)  A %s  sum %.4f' % ('$name', d['value'], d['ms_per_step'], d['ms_per_step_median'], ' '.join('%.1f' % (1e3*v) for v in r['per_iteration_ms']), r['atrous_x5_ms']))"
done
