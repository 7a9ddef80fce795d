root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
for how in "" 2; do echo "GF2_RREF_BCAST=$how"; GF2_RREF_BCAST=$how timeout -k 10 300 python3 profiles/time_rref_small.py 2>&1 | head -7; done > $out/rref_small_stage.log 2>&1; cat $out/rref_small_stage.log
