#!/bin/bash
# the record sampler with its lanes' records turned in LDS (no 16-way bank conflict of the slot stores): parity, then gf2_mc_run end to
# end at 7..9 wavefronts per CU
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "mc or monte or records or misfit" > gpurun_out/r04/mc10_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r04/mc10_tests.log
tail -3 gpurun_out/r04/mc10_tests.log
GF2_SAMPLER_WAVES="7 8 9 7 8 9" python3 profiles/time_mc.py > gpurun_out/r04/mc_turned_time.log 2>&1
cat gpurun_out/r04/mc_turned_time.log
