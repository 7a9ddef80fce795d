root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
python3 profiles/ab_inprocess.py --rounds 6 --steps 10 f=0 f=0,o9=1 f=0,o9=2 f=0,o9=3 > $out/ab4.log 2>&1; tail -5 $out/ab4.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "three_routes or hashed" > $out/fuzz_new.log 2>&1; echo "rc=$?" >> $out/fuzz_new.log; tail -4 $out/fuzz_new.log
