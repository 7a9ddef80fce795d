#!/bin/bash
# FETCH_SIZE of the one-stream bench run, three times: how much the gather kernel's fetch varies from run to run
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24"
for i in 1 2 3; do
  rm -rf /tmp/pmc_rep_$i
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_rep_$i -- python3 $root/bench.py $A --one-stream > /dev/null 2>&1 || exit 1
  python3 $root/profiles/pmc_summary.py $(find /tmp/pmc_rep_$i -name '*counter_collection.csv') | grep -E "slab_(gather|compact)" | cut -c1-140
done > $out/traffic_repeat.txt
cat $out/traffic_repeat.txt
