#!/bin/bash
# GPU box: the full -m gpu suite, then the launch-boundary probe (trace build) under rocprofv3 with the non-temporal stores on and off
set -e
R=$(pwd); OUT=$R/gpurun_out/${1:-r4d}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for nt in 1 0; do
  RMD_NT_OUT=$nt RMD_LIB_PATH=$R/build/variants/librmd_trace.so PROBE_OUT=$OUT/spans_nt$nt.json timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/bprof_nt$nt -o r -- python3 $R/tools/boundary_probe.py > $OUT/boundary_nt$nt.log 2>&1 || { tail -5 $OUT/boundary_nt$nt.log; exit 1; }
  python3 $R/tools/boundary_probe.py --reduce $OUT/bprof_nt$nt $OUT/spans_nt$nt.json | tee $OUT/boundary_nt$nt.txt
done
