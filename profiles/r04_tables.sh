root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "rref or small" > $out/rref_tests.log 2>&1; echo "rref rc=$?" >> $out/rref_tests.log
tail -4 $out/rref_tests.log
timeout -k 10 300 python3 profiles/time_rref_small.py > $out/rref_small.log 2>&1; cat $out/rref_small.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_tables.py -x -q -m gpu --durations=8 > $out/tables_tests.log 2>&1; echo "tables rc=$?" >> $out/tables_tests.log
tail -25 $out/tables_tests.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slab or config4_code_itself or redo or syndrome_table" > $out/syn_tests2.log 2>&1; echo "tests rc=$?" >> $out/syn_tests2.log
tail -5 $out/syn_tests2.log
