#!/bin/bash
# record sampler with leftovers: parity (caps 0 / 2 / 4 / 6 / 8 against the oracle), then gf2_mc_run end to end per cap
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "record_sampler_other_shapes or dense_and_sparse_pipelines" > gpurun_out/r04/tails_tests.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r04/tails_tests.log
tail -5 gpurun_out/r04/tails_tests.log
[ $rc -eq 0 ] || exit 1
GF2_TAIL_CAP="0 8 6 0 8 6 4" python3 profiles/time_mc.py > gpurun_out/r04/tails_time.log 2>&1
cat gpurun_out/r04/tails_time.log
