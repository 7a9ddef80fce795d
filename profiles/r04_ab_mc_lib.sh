#!/bin/bash
# Same-box A/B of gf2_mc_run end to end: the working tree's library against scratch_ab/<name>.so, alternating.
#   bash profiles/r04_ab_mc_lib.sh <name> [rounds]
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
use() { if [ $1 = new ]; then cp /tmp/new.so $root/quantum_css_codes_amd/libgf2hip.so; else cp $root/scratch_ab/$1.so $root/quantum_css_codes_amd/libgf2hip.so; fi; }
: > $out/ab_mc_lib.txt
use new
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "mc or monte or records or misfit" > $out/ab_mc_tests.log 2>&1
echo "tests rc=$?" | tee -a $out/ab_mc_tests.log
for i in $(seq 1 ${2:-4}); do
  for w in new $1; do
    use $w
    python3 profiles/time_mc.py 2>/dev/null | tail -1 | sed "s/^/$w /" | tee -a $out/ab_mc_lib.txt
  done
done
use new
