root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 profiles/r05_rref_dev.py tall > $out/tall5.log 2>&1 || { tail -20 $out/tall5.log; exit 1; }
tail -1 $out/tall5.log
{ python3 profiles/r05_rref_one.py 32768 65536 1 -1; python3 profiles/r05_rref_one.py 8192 16384 4 -1; python3 profiles/r05_rref_one.py 16384 32768 1 -1; } > $out/big5.log 2>&1
cat $out/big5.log
python -m pytest tests -m gpu -q -x -k "tall or streamed or 8192 or 256_mib or look_ahead" > $out/big5_tests.log 2>&1; rc=$?; tail -3 $out/big5_tests.log; exit $rc
