root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 profiles/r05_rref_dev.py tall > $out/tall11.log 2>&1 || { tail -20 $out/tall11.log; exit 1; }
tail -1 $out/tall11.log
for i in 1 2; do python3 profiles/r05_rref_one.py 32768 65536 1 -1; python3 profiles/r05_rref_one.py 8192 16384 4 -1; python3 profiles/r05_rref_one.py 16384 32768 1 -1; python3 profiles/r05_rref_one.py 8192 8192 1 -1; python3 profiles/r05_rref_one.py 5000 20000 8 -1; done > $out/big11.log 2>&1
cat $out/big11.log
python -m pytest tests -m gpu -q -x -k "tall or streamed or 8192 or 256_mib or look_ahead or fuzz" > $out/big11_tests.log 2>&1; rc=$?; tail -3 $out/big11_tests.log; exit $rc
