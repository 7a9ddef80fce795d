# rocprofv3 kernel trace of gf2_mc_run at n = 4096 (profiles/time_mc.py); run from the repo root on the GPU box
root=$(pwd); mkdir -p $root/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/r02/mc_trace
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/r02/mc_trace -- python3 $root/profiles/time_mc.py > $root/gpurun_out/r02/mc_trace.log 2>&1 || exit 1
python3 $root/profiles/summarize.py $(find $root/gpurun_out/r02/mc_trace -name '*kernel_trace.csv') > $root/gpurun_out/r02/mc_trace.md
tail -1 $root/gpurun_out/r02/mc_trace.log
