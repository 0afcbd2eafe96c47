#!/bin/bash
# GPU box: tools/box_probe.py (CROSS / WAVELET lines) once per library given, under rocprofv3 kernel stats
# usage: tools/r3_wab.sh <outname> <lib or "default"> ...
R=$(pwd); OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset RMD_LIB_PATH; else export RMD_LIB_PATH=$R/$lib; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o r -- python3 $R/tools/box_probe.py > $OUT/probe_$name.log 2>&1 || { tail -5 $OUT/probe_$name.log; exit 1; }
  echo "== $name"; grep -E "CROSS|WAVELET" $OUT/probe_$name.log
  python3 $R/tools/kstats.py $OUT/prof_$name weighted | cut -c1-130
done
