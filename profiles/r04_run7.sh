root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "rref or small" > $out/rref_tests.log 2>&1; echo "rref rc=$?" >> $out/rref_tests.log
tail -3 $out/rref_tests.log
for how in "" 0 1; do echo "GF2_RREF_BCAST=$how"; GF2_RREF_BCAST=$how timeout -k 10 300 python3 profiles/time_rref_small.py 2>&1 | head -5; done > $out/rref_small_hybrid.log 2>&1; cat $out/rref_small_hybrid.log
