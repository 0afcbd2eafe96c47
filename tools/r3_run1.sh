#!/bin/bash
# GPU box: round-3 baseline -- tests (product + experiments build), bench, kernel stats + trace
set -e
R=$(pwd); OUT=$R/gpurun_out/r3a; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
RMD_LIB_PATH=$R/build/variants/librmd_experiments.so timeout -k 10 400 python -m pytest tests -m gpu -x -q > $OUT/tests_exp.log 2>&1 || { tail -30 $OUT/tests_exp.log; exit 1; }
tail -3 $OUT/tests_exp.log
tools/final_profile.sh $OUT/prof
python3 tools/frame_gaps.py $OUT/prof/prof > $OUT/gaps.txt 2>&1 || true
cat $OUT/gaps.txt
