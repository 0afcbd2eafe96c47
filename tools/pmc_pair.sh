#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_pair.sh <outdir> [variants...] -- SQ counter passes of
# tools/atrous_probe2.py (one rocprofv3 run per pass; counters only with --kernel-trace)
set -e
OUT=$(realpath -m "$1"); shift; R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT" -o pass$i -- python3 "$R/tools/atrous_probe2.py" "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
ls "$OUT"
