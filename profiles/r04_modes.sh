#!/bin/bash
mkdir -p gpurun_out/r04
for p in 1 2 3; do echo "process $p"; python3 profiles/r04_modes.py --sides 5 --rounds 2 --steps 10; done > gpurun_out/r04/modes.log 2>&1
cat gpurun_out/r04/modes.log
