#!/bin/bash
mkdir -p gpurun_out/r04
for p in 1 2; do echo "process $p"; timeout -k 10 300 python3 profiles/r04_place.py; done > gpurun_out/r04/place.log 2>&1
cat gpurun_out/r04/place.log
