#!/bin/bash
# GPU box: the bench line, then rocprofv3 kernel traces of (a) the fused one-call GBuffer loop, (b) the eight-call chain, (c) bench.py
set -e
R=$(pwd); OUT=$R/gpurun_out/${1:-r4c}; mkdir -p $OUT
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
tail -c 3000 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
for form in fused chain; do
  PROBE_FORM=$form timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$form -o r -- python3 $R/tools/gbuffer_probe.py > $OUT/probe_$form.log 2>&1 || { tail -5 $OUT/probe_$form.log; exit 1; }
  tail -1 $OUT/probe_$form.log
  find $OUT/prof_$form -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$form.csv \;
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o r -- python3 $R/bench.py --no-cpu-baseline --no-other-sizes > $OUT/prof_bench.log 2>&1 || { tail -5 $OUT/prof_bench.log; exit 1; }
find $OUT/prof_bench -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_bench.csv \;
cd $R
python3 tools/frame_gaps.py $OUT/prof_bench --ramp > $OUT/gaps_bench.txt 2>&1 || true
python3 tools/frame_gaps.py $OUT/prof_fused --ramp > $OUT/gaps_fused.txt 2>&1 || true
head -20 $OUT/gaps_bench.txt
