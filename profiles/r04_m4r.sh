#!/bin/bash
# the wavefront-per-matrix RREF kernel, four pivots at a time: parity (every rref test), then 256 MiB batches per variant
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "rref" > gpurun_out/r04/m4r_tests.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r04/m4r_tests.log; tail -4 gpurun_out/r04/m4r_tests.log
[ $rc -eq 0 ] || exit 1
: > gpurun_out/r04/m4r_time.log
for v in 2 ""; do
  echo "GF2_RREF_BCAST=$v" >> gpurun_out/r04/m4r_time.log
  GF2_RREF_BCAST=$v python3 profiles/time_rref_small.py >> gpurun_out/r04/m4r_time.log 2>&1
done
cat gpurun_out/r04/m4r_time.log
