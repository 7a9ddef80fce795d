#!/bin/bash
# PMC passes over any timing script (one rocprofv3 run per counter group); run from the repo root on the GPU box:
#   bash profiles/pmc_any.sh <tag> profiles/time_sampler.py [args]   -> gpurun_out/pmc_<tag>_<k>/ and gpurun_out/pmc_<tag>.md
tag=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
k=0
for group in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM"; do
    k=$((k + 1))
    rm -rf $root/gpurun_out/pmc_${tag}_$k
    rocprofv3 --pmc $group --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_$k -- python3 $root/$1 ${@:2} > $root/gpurun_out/pmc_${tag}_$k.log 2>&1 || exit 1
done
cd $root
python3 profiles/pmc_summary.py $(find gpurun_out/pmc_${tag}_* -name '*counter_collection.csv') > gpurun_out/pmc_${tag}.md
