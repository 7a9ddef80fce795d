#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 500 python3 profiles/time_hashed.py > gpurun_out/r04/hashed_time.log 2>&1
cat gpurun_out/r04/hashed_time.log
