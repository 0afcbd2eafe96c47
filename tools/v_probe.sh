#!/bin/bash
# usage (on the GPU box, from the repo root): tools/v_probe.sh <outdir> [RMD_V_WORKGROUPS values...]
# rocprofv3 kernel stats of bench.py for the in-frame T and V launches under different V grids
set -e
OUT=$(realpath -m "$1"); shift; R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  RMD_V_WORKGROUPS=$n timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/v$n" -o r -- python3 "$R/bench.py" --no-cpu-baseline --no-other-sizes --steps 30 --warmup 6 > "$OUT/v$n.log" 2>&1 || { echo "run $n failed"; tail -5 "$OUT/v$n.log"; exit 1; }
  echo "== RMD_V_WORKGROUPS=$n"
  grep '"metric"' "$OUT/v$n.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('Mpix/s', d['value'], 'ms/frame', d['ms_per_step'])"
  find "$OUT/v$n" -name '*kernel_stats.csv' | xargs grep -h "variance\|temporal" | cut -c1-150
done
