root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
bash profiles/r04_ab_opts.sh "--slab-pass-log2 22|--slab-pass-log2 23|--slab-pass-log2 24|--slab-pass-log2 21|--one-stream|--one-stream --slab-pass-log2 24" 3
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "rref or small" > $out/rref_tests.log 2>&1; echo "rref rc=$?" >> $out/rref_tests.log
tail -3 $out/rref_tests.log
timeout -k 10 300 python3 profiles/time_rref_small.py > $out/rref_small_lds.log 2>&1; head -3 $out/rref_small_lds.log
GF2_RREF_BCAST=1 timeout -k 10 300 python3 profiles/time_rref_small.py > $out/rref_small_readlane.log 2>&1; head -3 $out/rref_small_readlane.log
