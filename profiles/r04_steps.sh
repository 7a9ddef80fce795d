#!/bin/bash
# does the length of the timed region matter (clocks under sustained load)?  default line at 5 / 10 / 20 / 40 / 80 steps, twice
mkdir -p gpurun_out/r04; : > gpurun_out/r04/steps.txt
for rep in 1 2; do
  for k in 5 10 20 40 80; do
    python3 bench.py --steps $k --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $k  frac %.4f  %.3f ms' % (d['roofline']['frac'], d['ms_per_step']))" | tee -a gpurun_out/r04/steps.txt
  done
done
