#!/bin/bash
# GPU box: -m gpu suite, bench line, kernel trace of bench (frame gaps), boundary probe
set -e
R=$(pwd); OUT=$R/gpurun_out/${1:-r4e}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'frac', d['roofline']['frac'], d['roofline']['per_iteration_ms'])
print('other', d['other_sizes'])
c=d['cornell_sequence_4k']; print('cornell', c['end_to_end_u8'], c['float_planes'])
print('cpu', d['cpu_baseline']['value'])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o r -- python3 $R/bench.py --no-cpu-baseline --no-other-sizes > $OUT/prof_bench.log 2>&1 || { tail -5 $OUT/prof_bench.log; exit 1; }
find $OUT/prof_bench -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_bench.csv \;
cd $R
python3 tools/frame_gaps.py $OUT/prof_bench --frames=128:177 | tee $OUT/gaps_bench.txt
cd /tmp
for nt in 1 0; do
  RMD_NT_OUT=$nt RMD_LIB_PATH=$R/build/variants/librmd_trace.so PROBE_OUT=$OUT/spans_nt$nt.json timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/bprof_nt$nt -o r -- python3 $R/tools/boundary_probe.py > $OUT/boundary_nt$nt.log 2>&1 || { tail -5 $OUT/boundary_nt$nt.log; exit 1; }
  python3 $R/tools/boundary_probe.py --reduce $OUT/bprof_nt$nt $OUT/spans_nt$nt.json | tee $OUT/boundary_nt$nt.txt
done
