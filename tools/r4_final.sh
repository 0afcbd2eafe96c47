#!/bin/bash
# GPU box, repo root: the round's numbers of record -> gpurun_out/<out>/ (copied into profiles/ by hand afterwards)
#   tests (product + experiments build), the bench line, rocprofv3 kernel stats + frame gaps of the same workload, kernel stats of the
#   one-call GBuffer loop and of the eight-call chain, PMC passes (a-trous probe; FETCH / WRITE of the frame loop), the in-kernel
#   clock and the launch-boundary probe (trace build), the 2-rank launcher-less bench over gloo (rehearsal of the N > 1 path)
set -e
R=$(pwd); OUT=$R/gpurun_out/${1:-r4final}; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
RMD_LIB_PATH=$R/build/variants/librmd_experiments.so timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_exp.log 2>&1 || { tail -40 $OUT/tests_exp.log; exit 1; }
tail -2 $OUT/tests_exp.log
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o r -- python3 $R/bench.py --no-cpu-baseline --no-other-sizes > $OUT/prof_bench.log 2>&1 || { tail -5 $OUT/prof_bench.log; exit 1; }
find $OUT/prof_bench -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_bench.csv \;
for form in fused chain; do
  PROBE_FORM=$form timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$form -o r -- python3 $R/tools/gbuffer_probe.py > $OUT/probe_$form.log 2>&1 || { tail -5 $OUT/probe_$form.log; exit 1; }
  grep "frames of" $OUT/probe_$form.log
  find $OUT/prof_$form -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_$form.csv \;
done
cd $R
python3 tools/frame_gaps.py $OUT/prof_bench --frames=186:235 > $OUT/gaps_bench.txt 2>&1 || true
python3 tools/frame_gaps.py $OUT/prof_bench --ramp > $OUT/ramp_bench.txt 2>&1 || true
python3 tools/frame_gaps.py $OUT/prof_fused > $OUT/gaps_fused.txt 2>&1 || true
python3 tools/frame_gaps.py $OUT/prof_chain > $OUT/gaps_chain.txt 2>&1 || true
head -12 $OUT/gaps_bench.txt
tools/pmc_passes.sh $OUT/pmc > $OUT/pmc_passes.log 2>&1 || true
tools/pmc_frame.sh $OUT/pmc_frame > $OUT/pmc_frame.txt 2>&1 || true
RMD_LIB_PATH=$R/build/variants/librmd_trace.so timeout -k 10 200 python3 tools/atrous_trace.py > $OUT/trace.txt 2>&1 || true
grep "shader clock\|kernel span" $OUT/trace.txt || true
python3 tools/pmc_traffic.py $OUT/pmc $OUT/pmc_traffic.json 2.04 $OUT/pmc_frame > $OUT/pmc_traffic.txt 2>&1 || true
python3 tools/pmc_summary.py $OUT/pmc > $OUT/pmc_atrous.txt 2>&1 || true
cd /tmp
for nt in 1 0; do
  RMD_NT_OUT=$nt RMD_LIB_PATH=$R/build/variants/librmd_trace.so PROBE_OUT=$OUT/spans_nt$nt.json timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/bprof_nt$nt -o r -- python3 $R/tools/boundary_probe.py > $OUT/boundary_nt$nt.log 2>&1 || { tail -5 $OUT/boundary_nt$nt.log; exit 1; }
  python3 $R/tools/boundary_probe.py --reduce $OUT/bprof_nt$nt $OUT/spans_nt$nt.json > $OUT/boundary_nt$nt.txt
done
cd $R
RMD_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 4 --no-other-sizes > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err || { tail -5 $OUT/bench_n2_gloo.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'frac', d['roofline']['frac'], d['roofline']['per_iteration_ms'])
print('other', d['other_sizes'])
for k in ('cornell_sequence_4k','cornell_1080p'):
    c=d[k]; print(k, c['end_to_end_u8'], c['float_planes'], c['pcie_inclusive_u8'])
print('cpu', d['cpu_baseline']['value'])
n=[json.loads(l) for l in open('$OUT/bench_n2_gloo.json') if l.startswith('{')][0]
print('N=2 (gloo, one GPU shared):', n['value'], n['ms_per_step'], n['config']['workload'][-80:], n['halo_bytes_per_frame_rank0'], n['roofline']['traffic'])
PY
