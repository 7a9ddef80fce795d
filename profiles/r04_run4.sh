root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "slab or config4 or redo or monte_carlo or sparse" > $out/slab_tests6.log 2>&1; echo "tests rc=$?" >> $out/slab_tests6.log
tail -4 $out/slab_tests6.log
python3 profiles/ab_inprocess.py --rounds 8 --steps 10 f=0 f=0x80000 f=0xC0000 f=0,o9=0 > $out/ab2.log 2>&1; tail -5 $out/ab2.log
python3 profiles/ab_inprocess.py --one-stream --rounds 5 --steps 10 f=0 f=0x80000 f=0xC0000 > $out/ab2_one.log 2>&1; tail -4 $out/ab2_one.log
