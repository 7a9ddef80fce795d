root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
for r in 32 40 48 32 40 48; do v=$(( (r + 1) * 256 )); python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 $v; done > $out/big12.log 2>&1
for r in 32 40 48; do v=$(( (r + 1) * 256 )); python3 profiles/r05_rref_one.py 8192 16384 4 -1 -1 $v; python3 profiles/r05_rref_one.py 16384 32768 1 -1 -1 $v; done >> $out/big12.log 2>&1
cat $out/big12.log
