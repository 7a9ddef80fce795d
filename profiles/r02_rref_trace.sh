# rocprofv3 kernel trace only of gf2_rref_batch_dev on the three shapes (quick look; r02_rref_prof.sh adds the PMC passes)
root=$(pwd); mkdir -p $root/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
for shape in "2048 4096 1" "2048 4096 256" "32768 65536 1"; do
  tag=$(echo $shape | tr ' ' 'x')
  rm -rf $root/gpurun_out/r02/rref_trace_$tag
  rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/r02/rref_trace_$tag -- python3 $root/profiles/time_rref.py $shape > $root/gpurun_out/r02/rref_trace_$tag.log 2>&1 || exit 1
  python3 $root/profiles/summarize.py $(find $root/gpurun_out/r02/rref_trace_$tag -name '*kernel_trace.csv') > $root/gpurun_out/r02/rref_trace_$tag.md
  tail -1 $root/gpurun_out/r02/rref_trace_$tag.log
done
