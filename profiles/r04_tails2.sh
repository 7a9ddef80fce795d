#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "record_sampler_other_shapes" > gpurun_out/r04/tails_tests.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r04/tails_tests.log
tail -3 gpurun_out/r04/tails_tests.log
[ $rc -eq 0 ] || exit 1
bash profiles/r04_ab_mc_libs.sh "new base" 4
GF2_TAIL_CAP="0 8 6" python3 profiles/time_mc.py 2>&1 | head -3
