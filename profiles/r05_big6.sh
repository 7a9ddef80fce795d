root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 profiles/r05_rref_dev.py tall > $out/tall6.log 2>&1 || { tail -20 $out/tall6.log; exit 1; }
tail -1 $out/tall6.log
for v in 0 2 4 0 2 4; do python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 $v; done > $out/big6.log 2>&1
for v in 0 2 4 0 2 4; do python3 profiles/r05_rref_one.py 8192 16384 4 -1 -1 $v; done >> $out/big6.log 2>&1
for v in 0 2 4; do python3 profiles/r05_rref_one.py 16384 32768 1 -1 -1 $v; done >> $out/big6.log 2>&1
for v in 0 2 4; do python3 profiles/r05_rref_one.py 8192 8192 1 -1 -1 $v; done >> $out/big6.log 2>&1
cat $out/big6.log
