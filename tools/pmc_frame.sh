#!/bin/bash
# usage (GPU box, repo root): tools/pmc_frame.sh <outdir> -- FETCH_SIZE / WRITE_SIZE of every kernel of the frame loop
# (two rocprofv3 passes over bench.py; counters only with --kernel-trace), reduced with tools/pmc_summary.py --all
set -e
OUT=$(realpath -m "$1"); R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT" -o pass$i -- python3 "$R/bench.py" --no-cpu-baseline --no-other-sizes --steps 20 --warmup 8 > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/pass$i.log"; }
done
python3 "$R/tools/pmc_summary.py" "$OUT" --all --median
