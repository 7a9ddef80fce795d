root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 profiles/r05_rref_dev.py check > $out/dev_check.log 2>&1 || { tail -20 $out/dev_check.log; exit 1; }
tail -2 $out/dev_check.log
{ python3 profiles/r05_rref_one.py 2048 4096 256 -1; python3 profiles/r05_rref_one.py 2048 4096 1 -1; python3 profiles/r05_rref_one.py 2048 4096 256 -1; python3 profiles/r05_rref_one.py 2048 4096 1 -1; python3 profiles/r05_rref_one.py 4096 8192 16 -1; python3 profiles/r05_rref_one.py 1024 2048 512 -1; } > $out/snap.log 2>&1
cat $out/snap.log
python -m pytest tests -m gpu -q -k "rref or normalize or nullspace or css_code or randomised" > $out/snap_tests.log 2>&1; rc=$?; tail -3 $out/snap_tests.log; exit $rc
