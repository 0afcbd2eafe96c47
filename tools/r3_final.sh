#!/bin/bash
# GPU box: the round's numbers of record -- full -m gpu suite, the bench line, rocprofv3 kernel stats of the same workload,
# frame gaps, and the 2-rank launcher-less bench (gloo, one GPU) as a rehearsal of the N > 1 path
set -e
R=$(pwd); OUT=$R/gpurun_out/${1:-r3final}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
tools/final_profile.sh $OUT/prof
python3 tools/frame_gaps.py $OUT/prof/prof > $OUT/gaps.txt 2>&1 || true
cat $OUT/gaps.txt
RMD_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 4 --no-other-sizes > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err || { tail -5 $OUT/bench_n2_gloo.err; exit 1; }
python3 -c "
import json
d=[json.loads(l) for l in open('$OUT/bench_n2_gloo.json') if l.startswith('{')][0]
print('N=2 (gloo, one GPU shared):', d['value'], d['ms_per_step'], d['config']['exchanges'], d['halo_bytes_per_frame_rank0'])"
