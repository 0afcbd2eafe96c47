#!/bin/bash
# GPU box: parity tests of the uchar4 filters, then tools/box_probe.py at 4K under rocprofv3 (kernel stats of every mode)
set -e
R=$(pwd); OUT=$R/gpurun_out/${1:-r3w}; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_weighted_filter.py tests/test_box_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r -- python3 $R/tools/box_probe.py > $OUT/probe.log 2>&1
cd $R; grep -v simple_timer $OUT/probe.log | tail -8
python3 tools/kstats.py $OUT/prof box gauss weighted
