# Final-kernel evidence for the headline path (run from the repo root on the GPU box):
#   1. rocprofv3 --kernel-trace of the default bench command (two streams) and of --one-stream -> per-kernel durations
#   2. HBM-side traffic of the slab pipeline: FETCH_SIZE and WRITE_SIZE in separate --pmc passes over the one-stream bench
#      (per-kernel counters need the kernels of one call to run alone)
# The program itself follows `--` (no env/bash hop).
root=$(pwd); out=$root/gpurun_out/r02; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 22"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_two -- python3 $root/bench.py $A > $out/trace_two.json 2> $out/trace_two.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_one -- python3 $root/bench.py $A --one-stream > $out/trace_one.json 2> $out/trace_one.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_one_$c -- python3 $root/bench.py $A --one-stream > $out/pmc_one_$c.json 2> $out/pmc_one_$c.err || exit 1
done
cd $root
python3 profiles/summarize.py $(find $out/trace_two -name '*kernel_trace.csv') > $out/trace_two.md
python3 profiles/summarize.py $(find $out/trace_one -name '*kernel_trace.csv') > $out/trace_one.md
python3 profiles/pmc_summary.py $(find $out/pmc_one_* -name '*counter_collection.csv') > $out/pmc_one.md
cp $(find $out/trace_one -name '*kernel_stats.csv' | head -1) $out/trace_one_kernel_stats.csv
cp $(find $out/trace_two -name '*kernel_stats.csv' | head -1) $out/trace_two_kernel_stats.csv
head -8 $out/trace_two.md; head -8 $out/trace_one.md; grep slab $out/pmc_one.md
