root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 profiles/r05_rref_dev.py tall > $out/tall7.log 2>&1 || { tail -20 $out/tall7.log; exit 1; }
tail -1 $out/tall7.log
for r in -1 4 8 16 24 40 56 -1; do v=$r; if [ $r -ge 0 ]; then v=$(( (r + 1) * 256 )); fi; python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 $v; done > $out/big7.log 2>&1
for r in -1 4 16 40 -1; do v=$r; if [ $r -ge 0 ]; then v=$(( (r + 1) * 256 )); fi; python3 profiles/r05_rref_one.py 8192 16384 4 -1 -1 $v; python3 profiles/r05_rref_one.py 16384 32768 1 -1 -1 $v; python3 profiles/r05_rref_one.py 8192 8192 1 -1 -1 $v; done >> $out/big7.log 2>&1
cat $out/big7.log
