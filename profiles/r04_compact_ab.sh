#!/bin/bash
# compact kernel variants: slab parity tests, then same-box A/B of the working tree's library against scratch_ab/base.so
mkdir -p gpurun_out/r04
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "slab or sparse or three_routes or syndrome" > gpurun_out/r04/compact_tests.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r04/compact_tests.log; tail -3 gpurun_out/r04/compact_tests.log
[ $rc -eq 0 ] || exit 1
bash profiles/r04_ab_lib.sh base 4 2>&1 | tail -12
