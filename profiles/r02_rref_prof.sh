# rocprofv3 kernel trace and HBM-side PMC traffic of gf2_rref_batch_dev on the three shapes bench.py reports (run from the repo root on the GPU box)
root=$(pwd); mkdir -p $root/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
for shape in "2048 4096 1" "2048 4096 256" "32768 65536 1"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/r02/rref_trace_$tag -- python3 $root/profiles/time_rref.py $shape > $root/gpurun_out/r02/rref_trace_$tag.log 2>&1 || exit 1
  python3 $root/profiles/summarize.py $(find $root/gpurun_out/r02/rref_trace_$tag -name '*kernel_trace.csv') > $root/gpurun_out/r02/rref_trace_$tag.md
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/gpurun_out/r02/rref_pmc_${tag}_$c -- python3 $root/profiles/time_rref.py $shape > $root/gpurun_out/r02/rref_pmc_${tag}_$c.log 2>&1 || exit 1
  done
  python3 $root/profiles/pmc_summary.py $(find $root/gpurun_out/r02/rref_pmc_${tag}_* -name '*counter_collection.csv') > $root/gpurun_out/r02/rref_pmc_$tag.md
  tail -2 $root/gpurun_out/r02/rref_trace_$tag.log
done
