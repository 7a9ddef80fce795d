root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 profiles/r05_rref_dev.py tall > $out/tall3.log 2>&1 || { tail -20 $out/tall3.log; exit 1; }
tail -1 $out/tall3.log
{ python3 profiles/r05_rref_one.py 32768 65536 1 -1; python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 1; python3 profiles/r05_rref_one.py 8192 16384 4 -1; python3 profiles/r05_rref_one.py 16384 32768 1 -1; } > $out/big3.log 2>&1
cat $out/big3.log
{ bash profiles/r05_diag.sh "2048 4096 256 4"; bash profiles/r05_diag.sh "2048 4096 1 2"; bash profiles/r05_diag.sh "2048 4096 1 4"; } > $out/diag3.log 2>&1
cat $out/diag3.log
