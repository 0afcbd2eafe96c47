#!/bin/bash
# GPU box: kernel stats of the frame loop at a size: tools/r3_kstats.sh <out> <W> <H>
R=$(pwd); OUT=$R/gpurun_out/${1:-r3k}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBE_W=${2:-3840} PROBE_H=${3:-2160} PROBE_FRAMES=64 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python3 $R/tools/frame_probe.py > $OUT/probe.log 2>&1
tail -1 $OUT/probe.log
cd $R; python3 tools/kstats.py $OUT/prof
