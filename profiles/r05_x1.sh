# one 2048 x 4096 matrix: rows per workgroup of the trailing pass (the pass is launch + table build + rows)
root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
for rw in -1 512 256 128 64 -1 128 64; do python3 profiles/r05_rref_one.py 2048 4096 1 -1 $rw; done > $out/x1_rows.log 2>&1
for rw in -1 128 64; do python3 profiles/r05_rref_one.py 2048 4096 1 4 $rw; python3 profiles/r05_rref_one.py 1024 2048 1 -1 $rw; python3 profiles/r05_rref_one.py 4096 8192 1 -1 $rw; python3 profiles/r05_rref_one.py 2048 4096 8 -1 $rw; done >> $out/x1_rows.log 2>&1
cat $out/x1_rows.log
