# streamed sweeps: parity on tall shapes, then the variants of the launcher (GF2_OPT_RREF_STREAM_VARIANT) on the 256 MiB matrix
root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 profiles/r05_rref_dev.py tall > $out/tall2.log 2>&1 || { tail -20 $out/tall2.log; exit 1; }
tail -1 $out/tall2.log
for v in 0 1 2 3; do python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 $v || exit 1; done > $out/big_variants.log 2>&1
for r in 16 24 32 48 56; do python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 $(( (r + 1) * 256 )) || exit 1; done >> $out/big_variants.log 2>&1
for v in 0 3; do python3 profiles/r05_rref_one.py 8192 16384 4 -1 -1 $v; python3 profiles/r05_rref_one.py 16384 32768 1 -1 -1 $v; done >> $out/big_variants.log 2>&1
cat $out/big_variants.log
