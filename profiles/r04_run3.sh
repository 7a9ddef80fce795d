root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slab or redo or config4_code" > $out/slab_tests5.log 2>&1; echo "tests rc=$?" >> $out/slab_tests5.log
tail -3 $out/slab_tests5.log
bash profiles/r04_ab_flags.sh "0 0x40000" 3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_new -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24 --one-stream > /dev/null 2>&1
python3 $root/profiles/summarize.py $(find /tmp/tr_new -name '*kernel_trace.csv') > $out/new_one_trace2.md
head -5 $out/new_one_trace2.md | cut -c1-150
