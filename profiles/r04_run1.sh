root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "slab or config4 or redo or monte_carlo or mc_ or sparse" > $out/slab_tests4.log 2>&1; echo "tests rc=$?" >> $out/slab_tests4.log
tail -6 $out/slab_tests4.log
bash profiles/r04_ab_flags.sh "0 0x40000" 4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_new -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24 --one-stream > /dev/null 2>&1
python3 $root/profiles/summarize.py $(find /tmp/tr_new -name '*kernel_trace.csv') > $out/new_one_trace.md
head -8 $out/new_one_trace.md | cut -c1-150
