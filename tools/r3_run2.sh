#!/bin/bash
# GPU box: full -m gpu suite, then the strip probe (8 ranks: exchange_iteration -1 / 3 / 2) with the RCCL loop-back timing
set -e
R=$(pwd); OUT=$R/gpurun_out/${1:-r3b}; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
PROBE_EXCHANGE="-1 3 2" PROBE_LOOPBACK=1 timeout -k 10 300 python3 tools/strip_probe.py 8 > $OUT/strip_probe_8.txt 2>&1 || { tail -20 $OUT/strip_probe_8.txt; exit 1; }
cat $OUT/strip_probe_8.txt
