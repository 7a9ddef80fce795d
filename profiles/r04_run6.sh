root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
python3 profiles/ab_inprocess.py --rounds 6 --steps 10 f=0 f=0,o2=0 f=0,o6=1 f=0,o7=2 f=0,o0=21 f=0x8000 > $out/ab3.log 2>&1; tail -7 $out/ab3.log
