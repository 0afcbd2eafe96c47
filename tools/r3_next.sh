#!/bin/bash
# GPU box (experiments build): tools/next_probe.py at 4K and 1080p, then the kernel stats of each form under rocprofv3
R=$(pwd); OUT=$R/gpurun_out/${1:-r3next}; mkdir -p $OUT
export RMD_LIB_PATH=$R/build/variants/librmd_experiments.so
timeout -k 10 300 python3 tools/next_probe.py > $OUT/probe4k.log 2>&1 || { tail -20 $OUT/probe4k.log; exit 1; }
grep next_probe $OUT/probe4k.log
PROBE_W=1920 PROBE_H=1080 PROBE_FRAMES=80 timeout -k 10 300 python3 tools/next_probe.py > $OUT/probe1080.log 2>&1 || { tail -20 $OUT/probe1080.log; exit 1; }
grep next_probe $OUT/probe1080.log
cd /tmp && export TMPDIR=/tmp
for mode in serial ahead; do
  PROBE_MODE=$mode timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$mode -o r -- python3 $R/tools/next_probe.py > $OUT/prof_$mode.log 2>&1
  echo "== $mode"; python3 $R/tools/kstats.py $OUT/prof_$mode atrous temporal variance | cut -c1-150
done
