#!/bin/bash
mkdir -p gpurun_out/r04; : > gpurun_out/r04/place2.log
for o in first last last first first last last first first last; do
  timeout -k 10 200 python3 profiles/r04_place2.py $o 2>/dev/null | tee -a gpurun_out/r04/place2.log
done
true
