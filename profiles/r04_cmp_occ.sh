#!/bin/bash
# compact kernel at 5 (kept), 4 and 3 workgroups per CU: one-stream kernel durations (is it waiting for latency?)
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
cd /tmp && export TMPDIR=/tmp
for w in new cmp4 cmp3; do
  if [ $w = new ]; then cp /tmp/new.so $root/quantum_css_codes_amd/libgf2hip.so; else cp $root/scratch_ab/$w.so $root/quantum_css_codes_amd/libgf2hip.so; fi
  rm -rf /tmp/tr_$w
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$w -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24 --one-stream > /dev/null 2>&1
  echo "one stream, $w:"; python3 $root/profiles/summarize.py $(find /tmp/tr_$w -name '*kernel_trace.csv') | grep -E "slab_(gather|compact)" | head -2 | cut -c1-150
done > $out/cmp_occ.log
cp /tmp/new.so $root/quantum_css_codes_amd/libgf2hip.so
cat $out/cmp_occ.log
