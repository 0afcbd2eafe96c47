#!/bin/bash
# GPU box: where a 1080p frame's time goes -- kernel trace of the frame loop (durations + gaps) and the per-workgroup timeline
R=$(pwd); OUT=$R/gpurun_out/${1:-r3c}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBE_W=${PW:-1920} PROBE_H=${PH:-1080} PROBE_FRAMES=80 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python3 $R/tools/frame_probe.py > $OUT/probe.log 2>&1
tail -2 $OUT/probe.log
cd $R
python3 tools/frame_gaps.py $OUT/prof | tee $OUT/gaps.txt
PROBE_W=${PW:-1920} PROBE_H=${PH:-1080} RMD_TRACE_PHASES=1 RMD_LIB_PATH=$R/build/variants/librmd_trace.so timeout -k 10 200 python3 tools/atrous_trace.py > $OUT/trace.txt 2>&1
grep -v "^ *$" $OUT/trace.txt | head -80
