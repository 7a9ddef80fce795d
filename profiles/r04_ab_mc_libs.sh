#!/bin/bash
# Same-box timing of gf2_mc_run end to end over several libraries (scratch_ab/<name>.so; "new" = the working tree's), alternating.
# No parity run: for what-if variants that time a wrong result.    bash profiles/r04_ab_mc_libs.sh "<names>" [rounds]
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
use() { if [ $1 = new ]; then cp /tmp/new.so $root/quantum_css_codes_amd/libgf2hip.so; else cp $root/scratch_ab/$1.so $root/quantum_css_codes_amd/libgf2hip.so; fi; }
: > $out/ab_mc_libs.txt
for i in $(seq 1 ${2:-3}); do
  for w in $1; do
    use $w
    python3 -O profiles/time_mc.py 2>&1 | tail -1 | sed "s/^/$w /" | tee -a $out/ab_mc_libs.txt
  done
done
use new
