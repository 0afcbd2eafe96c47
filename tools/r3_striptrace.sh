#!/bin/bash
# GPU box: kernel trace of one interior rank's strip of 8 (exchange_iteration 3): duration of every launch of a frame
R=$(pwd); OUT=$R/gpurun_out/${1:-r3st}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROBE_EXCHANGE="${2:-3}" timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o r -- python3 $R/tools/strip_probe.py 8 4 > $OUT/probe.log 2>&1
cd $R; grep "rank" $OUT/probe.log
python3 - <<PY
import csv, collections
rows=[]
for r in csv.DictReader(open('$OUT/prof/r_kernel_trace.csv')):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rmd::","").replace("rmd::",""), int(r["Grid_Size_X"])))
rows.sort()
# last 40% of the dispatches: the strip loop; group by (kernel, grid)
tail=rows[int(len(rows)*0.6):]
acc=collections.defaultdict(list)
for s,e,n,g in tail: acc[(n,g)].append((e-s)/1e3)
for (n,g),v in sorted(acc.items(), key=lambda kv: kv[0]):
    v.sort(); print(f"{n[:44]:44s} grid {g:8d}  calls {len(v):4d}  median {v[len(v)//2]:7.1f} us")
PY
