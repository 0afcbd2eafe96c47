#!/bin/bash
# GPU box: svgf parity tests, then bench.py's Cornell sequence + headline once per library given (A/B in one call)
R=$(pwd); OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
timeout -k 10 600 python -m pytest tests/test_svgf_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for rep in 1 2; do for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset RMD_LIB_PATH; else export RMD_LIB_PATH=$R/$lib; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { tail -5 $OUT/bench_$name.err; exit 1; }
  python3 -c "
import json
d=[json.loads(l) for l in open('$OUT/bench_$name.json') if l.startswith('{')][0]
c=d['cornell_sequence_4k']
print('%-14s 4K synthetic %8.1f Mpix/s %.4f ms | Cornell sequence %7.1f fps %.4f ms/frame | A %s' % ('$name', d['value'], d['ms_per_step'], c['fps'], c['ms_per_frame'], ' '.join('%.1f' % (1e3*v) for v in d['roofline']['per_iteration_ms'])))"
done; done
