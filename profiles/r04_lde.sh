#!/bin/bash
mkdir -p gpurun_out/r04
for o in 64,72 72,64 64,72 72,64; do echo "process $o"; timeout -k 10 300 python3 profiles/r04_lde.py $o 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r04/lde.log 2>&1
cat gpurun_out/r04/lde.log
