#!/bin/bash
# GPU box: PMC passes of the a-trous probe + the in-kernel clock (trace build) -> gpurun_out/<out>/
R=$(pwd); OUT=$R/gpurun_out/${1:-r3pmc}; mkdir -p $OUT
tools/pmc_passes.sh $OUT > $OUT/passes.log 2>&1; tail -3 $OUT/passes.log
RMD_LIB_PATH=$R/build/variants/librmd_trace.so timeout -k 10 200 python3 tools/atrous_trace.py > $OUT/trace.txt 2>&1
grep "shader clock\|kernel span" $OUT/trace.txt
ls $OUT | head -30
