#!/bin/bash
mkdir -p gpurun_out/r04; : > gpurun_out/r04/modes2.log
for rep in 1 2 3; do
  for k in 0 1 2 3 4; do
    python3 profiles/r04_modes2.py --dummies $k 2>/dev/null | tee -a gpurun_out/r04/modes2.log
    [ $k -gt 0 ] && python3 profiles/r04_modes2.py --dummies $k --close 2>/dev/null | tee -a gpurun_out/r04/modes2.log
  done
done
true
