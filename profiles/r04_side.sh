#!/bin/bash
# the side stream picked among five against the first one created (--side-candidates 1), alternating bench runs
mkdir -p gpurun_out/r04; : > gpurun_out/r04/side.txt
for rep in 1 2 3 4; do
  for k in 5k 5 1; do
    keep=""; n=$k; if [ $k = 5k ]; then keep=1; n=5; fi
    BENCH_KEEP_CANDIDATES=$keep python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --side-candidates $n 2>gpurun_out/r04/side.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('candidates $k  frac %.4f  %.3f ms  %s' % (d['roofline']['frac'], d['ms_per_step'], (d['config']['side_stream'] or {}).get('calibration_ms')))" | tee -a gpurun_out/r04/side.txt
  done
done
