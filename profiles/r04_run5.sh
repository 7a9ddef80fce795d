root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "monte_carlo or mc_ or sampler or config4" > $out/mc_tests.log 2>&1; echo "tests rc=$?" >> $out/mc_tests.log
tail -4 $out/mc_tests.log
python3 profiles/time_mc.py > $out/mc_time_new.log 2>&1; cat $out/mc_time_new.log
GF2_FLAGS=0x40000 python3 profiles/time_mc.py > $out/mc_time_sep.log 2>&1; cat $out/mc_time_sep.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_mc -- python3 $root/profiles/time_mc.py > /dev/null 2>&1
python3 $root/profiles/summarize.py $(find /tmp/tr_mc -name '*kernel_trace.csv') > $out/mc_trace.md
head -9 $out/mc_trace.md | cut -c1-150
