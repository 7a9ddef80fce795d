root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
for pad in -1 40 -1 40 90; do python3 profiles/r05_rref_one.py 2048 4096 256 4 -1 $pad || exit 1; done > $out/pad.log 2>&1
for pad in -1 40; do python3 profiles/r05_rref_one.py 2048 4096 64 4 -1 $pad || exit 1; done >> $out/pad.log 2>&1
cat $out/pad.log
