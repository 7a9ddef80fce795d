root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_tables.py -x -q -m gpu --durations=8 > $out/tables_tests.log 2>&1; echo "tables rc=$?" >> $out/tables_tests.log
tail -25 $out/tables_tests.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "syndrome_table or mid_size" --durations=5 > $out/syn_tests3.log 2>&1; echo "tests rc=$?" >> $out/syn_tests3.log
tail -12 $out/syn_tests3.log
