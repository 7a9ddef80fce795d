# Round 5: profiler evidence at the final kernels, all from one snapshot of the sources (their digest, profiles/csrc_digest.py, is
# written next to the numbers and into profiles/traffic.json; bench.py compares it at run time).  Run from the repo root on the
# GPU box:  bash profiles/r05_evidence.sh   -> gpurun_out/r05/ev_*   then, here:  python3 profiles/r05_collect.py
#   1. rocprofv3 --kernel-trace of the default bench (two streams) and of --one-stream           -> per-kernel durations
#   2. FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of the one-stream bench                   -> HBM-side traffic of the slab pipeline
#   3. kernel trace + FETCH_SIZE / WRITE_SIZE of gf2_rref_batch_dev on bench.py's three shapes   -> the sweep kernels (round 5)
#   4. kernel trace of gf2_mc_run at n = 4096
# The program itself follows `--` (no env / bash hop between rocprofv3 and python3); counters never together with other trace domains.
root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 $root/profiles/csrc_digest.py > $out/ev_csrc_digest.txt
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ev_trace_two -- python3 $root/bench.py $A > $out/ev_trace_two.json 2> $out/ev_trace_two.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ev_trace_one -- python3 $root/bench.py $A --one-stream > $out/ev_trace_one.json 2> $out/ev_trace_one.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/ev_pmc_one_$c -- python3 $root/bench.py $A --one-stream > $out/ev_pmc_one_$c.json 2> $out/ev_pmc_one_$c.err || exit 1
done
for shape in "2048 4096 1" "2048 4096 256" "32768 65536 1"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --kernel-trace --output-format csv -d $out/ev_rref_trace_$tag -- python3 $root/profiles/time_rref.py $shape > $out/ev_rref_trace_$tag.log 2>&1 || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/ev_rref_pmc_${tag}_$c -- python3 $root/profiles/time_rref.py $shape > $out/ev_rref_pmc_${tag}_$c.log 2>&1 || exit 1
  done
done
rocprofv3 --kernel-trace --output-format csv -d $out/ev_mc_trace -- python3 $root/profiles/time_mc.py > $out/ev_mc_trace.log 2>&1 || exit 1
cd $root
python3 profiles/summarize.py $(find $out/ev_trace_two -name '*kernel_trace.csv') > $out/ev_two_stream_kernel_trace.md
python3 profiles/summarize.py $(find $out/ev_trace_one -name '*kernel_trace.csv') > $out/ev_one_stream_kernel_trace.md
python3 profiles/pmc_summary.py $(find $out/ev_pmc_one_* -name '*counter_collection.csv') > $out/ev_slab_pipeline_pmc.md
for shape in 2048x4096x1 2048x4096x256 32768x65536x1; do
  python3 profiles/summarize.py $(find $out/ev_rref_trace_$shape -name '*kernel_trace.csv') > $out/ev_rref_${shape}_kernel_trace.md
  python3 profiles/pmc_summary.py $(find $out/ev_rref_pmc_${shape}_* -name '*counter_collection.csv') > $out/ev_rref_${shape}_pmc.md
  tail -n 1 $out/ev_rref_trace_$shape.log
done
python3 profiles/summarize.py $(find $out/ev_mc_trace -name '*kernel_trace.csv') > $out/ev_mc_trace.md
tail -n 1 $out/ev_mc_trace.log
head -8 $out/ev_two_stream_kernel_trace.md | cut -c1-150
for shape in 2048x4096x256 32768x65536x1; do head -6 $out/ev_rref_${shape}_kernel_trace.md | cut -c1-150; done
