#!/bin/bash
# GPU box: VALU / LDS counters of the uchar4 filter kernels (tools/box_probe.py), one rocprofv3 run per counter pass
R=$(pwd); OUT=$R/gpurun_out/${1:-r3wpmc}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT -o pass$i -- python3 $R/tools/box_probe.py > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
done
cd $R; python3 tools/pmc_summary.py $OUT --all > $OUT/summary.txt; grep -A14 "weighted_tile_kernel<true, true, false, 1, 4>" $OUT/summary.txt
