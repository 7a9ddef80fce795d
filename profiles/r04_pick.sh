#!/bin/bash
# bench.py with two sets of resident errors to choose from (default) against --one-region, alternating processes
mkdir -p gpurun_out/r04; : > gpurun_out/r04/pick.txt
for rep in 1 2 3 4 5; do
  for v in "" "--one-region"; do
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $v 2>gpurun_out/r04/pick.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['config'].get('resident_errors') or {}
print('%-14s frac %.4f  %.3f ms  %s' % ('${v:-two sets}', d['roofline']['frac'], d['ms_per_step'], {k: p[k] for k in ('first_ms','second_ms','picked') if k in p}))" | tee -a gpurun_out/r04/pick.txt
  done
done
tail -2 gpurun_out/r04/pick.err
