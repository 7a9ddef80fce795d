root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 -1 0x10000 > $out/big8.log 2>&1
cat $out/big8.log
bash profiles/r05_trace.sh "32768 65536 1 -1 -1 -1 0x10000" big_noahead > /dev/null || exit 1
python3 profiles/r05_timeline.py $(find $out/tr_big_noahead -name '*kernel_trace.csv') 60 > $out/big_noahead_timeline.txt
bash profiles/r05_trace.sh "32768 65536 1 -1 -1 1280" big_r4 > /dev/null || exit 1
python3 profiles/r05_timeline.py $(find $out/tr_big_r4 -name '*kernel_trace.csv') 60 > $out/big_r4_timeline.txt
