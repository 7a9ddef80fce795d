# kernel trace of the 256 MiB matrix at HEAD and the per-sweep periods (every 8th sweep)
root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
bash profiles/r05_trace.sh "32768 65536 1 -1" big_head3 > /dev/null || exit 1
python3 profiles/r05_timeline.py $(find $out/tr_big_head3 -name '*kernel_trace.csv') 100000 > $out/big_head3_timeline.txt
tail -1 $out/big_head3_timeline.txt
