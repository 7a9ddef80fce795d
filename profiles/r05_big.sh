# the 32768 x 65536 matrix at HEAD: times of the three routes (no profiler), then a kernel trace of the default and its timeline
root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
for k in -1 0; do python3 profiles/r05_rref_one.py 32768 65536 1 $k || exit 1; done > $out/big_times.log 2>&1
python3 profiles/r05_rref_one.py 8192 16384 4 -1 >> $out/big_times.log 2>&1 || exit 1
python3 profiles/r05_rref_one.py 8192 16384 4 0 >> $out/big_times.log 2>&1 || exit 1
python3 profiles/r05_rref_one.py 16384 32768 1 -1 >> $out/big_times.log 2>&1 || exit 1
python3 profiles/r05_rref_one.py 16384 32768 1 0 >> $out/big_times.log 2>&1 || exit 1
cat $out/big_times.log
bash profiles/r05_trace.sh "32768 65536 1 -1" big_head > /dev/null || exit 1
python3 profiles/r05_timeline.py $(find $out/tr_big_head -name '*kernel_trace.csv') 120 > $out/big_head_timeline.txt
cat $out/tr_big_head.md
