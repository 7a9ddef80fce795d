#!/bin/bash
# Instruction-cache counters of a timing script (one rocprofv3 run per group):  bash profiles/pmc_icache.sh <tag> profiles/time_slabs.py
tag=$1; shift
root=$(pwd)
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $root/gpurun_out/counters_avail.txt 2>&1
k=0
for group in "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
             "SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_VALU"; do
    k=$((k + 1))
    rm -rf $root/gpurun_out/pmc_${tag}_$k
    rocprofv3 --pmc $group --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_$k -- python3 $root/$1 ${@:2} > $root/gpurun_out/pmc_${tag}_$k.log 2>&1 || echo "group $k failed"
done
cd $root
python3 profiles/pmc_summary.py $(find gpurun_out/pmc_${tag}_* -name '*counter_collection.csv') > gpurun_out/pmc_${tag}.md
