#!/bin/bash
# usage (GPU box, repo root): tools/final_profile.sh <outdir> -- the bench line and the rocprofv3 kernel stats of the same workload
set -e
OUT=$(realpath -m "$1"); R=$(pwd)
mkdir -p "$OUT"
timeout -k 10 500 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o r -- python3 "$R/bench.py" --no-cpu-baseline --no-other-sizes > "$OUT/prof.log" 2>&1 || { tail -5 "$OUT/prof.log"; exit 1; }
cp "$OUT"/prof/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || find "$OUT/prof" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
tail -c 1500 "$OUT/bench.json"
