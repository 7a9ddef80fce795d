#!/bin/bash
# HBM-side traffic per kernel of profiles/time_slabs.py: FETCH_SIZE and WRITE_SIZE in separate passes (run from the repo root on the GPU box)
tag=${1:-traffic}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_$c -- python3 $root/profiles/time_slabs.py > $root/gpurun_out/pmc_${tag}_$c.log 2>&1 || exit 1
done
cd $root
python3 profiles/pmc_summary.py $(find gpurun_out/pmc_${tag}_* -name '*counter_collection.csv') > gpurun_out/pmc_${tag}.md
