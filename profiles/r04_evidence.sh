# Round 4: profiler evidence for the final kernels, all from one checkout (the commit is written next to the numbers).  Run from
# the repo root on the GPU box:  bash profiles/r04_evidence.sh   -> gpurun_out/r04/ev_*  (copy the .md / .json files into profiles/)
#   1. rocprofv3 --kernel-trace of the default bench (two streams) and of --one-stream       -> per-kernel durations
#   2. FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of the one-stream bench and of --algo dense -> HBM-side traffic per kernel
#   3. kernel trace + FETCH_SIZE / WRITE_SIZE of gf2_rref_batch_dev on bench.py's three shapes
#   4. kernel trace of gf2_mc_run at n = 4096
#   5., 6. see below
# The program itself follows `--` (no env / bash hop between rocprofv3 and python3).
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
git -C $root rev-parse HEAD > $out/ev_commit.txt 2>/dev/null || echo "(no git on the box: see the commit that holds this file)" > $out/ev_commit.txt
cd /tmp && export TMPDIR=/tmp
# (2^24 samples per step = four passes per call: the combine step of three of them rides in the next gather launch, as in the
# default run of 32 passes per call; a call's last pass is combined by a launch of its own)
A="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ev_trace_two -- python3 $root/bench.py $A > $out/ev_trace_two.json 2> $out/ev_trace_two.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ev_trace_one -- python3 $root/bench.py $A --one-stream > $out/ev_trace_one.json 2> $out/ev_trace_one.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/ev_pmc_one_$c -- python3 $root/bench.py $A --one-stream > $out/ev_pmc_one_$c.json 2> $out/ev_pmc_one_$c.err || exit 1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/ev_pmc_dense_$c -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 20 --algo dense > $out/ev_pmc_dense_$c.json 2> $out/ev_pmc_dense_$c.err || exit 1
done
for shape in "2048 4096 1" "2048 4096 256" "32768 65536 1"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --kernel-trace --output-format csv -d $out/ev_rref_trace_$tag -- python3 $root/profiles/time_rref.py $shape > $out/ev_rref_trace_$tag.log 2>&1 || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/ev_rref_pmc_${tag}_$c -- python3 $root/profiles/time_rref.py $shape > $out/ev_rref_pmc_${tag}_$c.log 2>&1 || exit 1
  done
done
rocprofv3 --kernel-trace --output-format csv -d $out/ev_mc_trace -- python3 $root/profiles/time_mc.py > $out/ev_mc_trace.log 2>&1 || exit 1
#   5. the small-matrix RREF kernel: trace + FETCH_SIZE / WRITE_SIZE over profiles/time_rref_small.py (VERDICT r03 item 4 asked for both)
rocprofv3 --kernel-trace --output-format csv -d $out/ev_rref_small_trace -- python3 $root/profiles/time_rref_small.py > $out/ev_rref_small_trace.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/ev_rref_small_pmc_$c -- python3 $root/profiles/time_rref_small.py > $out/ev_rref_small_pmc_$c.log 2>&1 || exit 1
done
#   6. the slab pipeline with the syndromes stored (secondary.read_write_1536B): trace + FETCH_SIZE / WRITE_SIZE over profiles/time_slabs_stored.py
rocprofv3 --kernel-trace --output-format csv -d $out/ev_stored_trace -- python3 $root/profiles/time_slabs_stored.py > $out/ev_stored_trace.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/ev_stored_pmc_$c -- python3 $root/profiles/time_slabs_stored.py > $out/ev_stored_pmc_$c.log 2>&1 || exit 1
done
cd $root
python3 profiles/summarize.py $(find $out/ev_trace_two -name '*kernel_trace.csv') > $out/ev_two_stream_kernel_trace.md
python3 profiles/summarize.py $(find $out/ev_trace_one -name '*kernel_trace.csv') > $out/ev_one_stream_kernel_trace.md
python3 profiles/pmc_summary.py $(find $out/ev_pmc_one_* -name '*counter_collection.csv') > $out/ev_slab_pipeline_pmc.md
python3 profiles/pmc_summary.py $(find $out/ev_pmc_dense_* -name '*counter_collection.csv') > $out/ev_dense_pmc.md
for shape in 2048x4096x1 2048x4096x256 32768x65536x1; do
  python3 profiles/summarize.py $(find $out/ev_rref_trace_$shape -name '*kernel_trace.csv') > $out/ev_rref_${shape}_kernel_trace.md
  python3 profiles/pmc_summary.py $(find $out/ev_rref_pmc_${shape}_* -name '*counter_collection.csv') > $out/ev_rref_${shape}_pmc.md
  tail -n 1 $out/ev_rref_trace_$shape.log
done
python3 profiles/summarize.py $(find $out/ev_mc_trace -name '*kernel_trace.csv') > $out/ev_mc_trace.md
python3 profiles/summarize.py $(find $out/ev_rref_small_trace -name '*kernel_trace.csv') > $out/ev_rref_small_trace.md
python3 profiles/pmc_summary.py $(find $out/ev_rref_small_pmc_* -name '*counter_collection.csv') > $out/ev_rref_small_pmc.md
python3 profiles/summarize.py $(find $out/ev_stored_trace -name '*kernel_trace.csv') > $out/ev_stored_trace.md
python3 profiles/pmc_summary.py $(find $out/ev_stored_pmc_* -name '*counter_collection.csv') > $out/ev_stored_pmc.md
cat $out/ev_rref_small_trace.log $out/ev_stored_trace.log
tail -n 1 $out/ev_mc_trace.log
head -8 $out/ev_two_stream_kernel_trace.md | cut -c1-150; head -8 $out/ev_one_stream_kernel_trace.md | cut -c1-150
grep -E "slab_|syndrome_tiled" $out/ev_slab_pipeline_pmc.md $out/ev_dense_pmc.md | cut -c1-170
