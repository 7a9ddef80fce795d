#!/bin/bash
# PMC passes over gf2_mc_run end to end (record sampler, gathers, misfits) + a kernel trace
mkdir -p gpurun_out/r04
bash profiles/pmc_any.sh mcrec4 profiles/time_mc.py
cp gpurun_out/pmc_mcrec4.md gpurun_out/r04/mc_pmc.md
root=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tr_mc
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_mc -- python3 $root/profiles/time_mc.py > /dev/null 2>&1
python3 $root/profiles/summarize.py $(find /tmp/tr_mc -name '*kernel_trace.csv') > $root/gpurun_out/r04/mc_trace.md
head -12 $root/gpurun_out/r04/mc_trace.md | cut -c1-160
