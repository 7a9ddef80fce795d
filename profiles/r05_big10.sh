root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
bash profiles/r05_trace.sh "32768 65536 1 -1 -1 2304" big_r8 > /dev/null || exit 1
python3 profiles/r05_timeline.py $(find $out/tr_big_r8 -name '*kernel_trace.csv') 400 > $out/big_r8_timeline.txt
