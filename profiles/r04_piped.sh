#!/bin/bash
# GF2_F_SLAB_PIPELINED (0x80000): compact kernels and gather kernels of a context on two streams, two record buffers
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "many_passes" > gpurun_out/r04/piped_tests.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r04/piped_tests.log; tail -3 gpurun_out/r04/piped_tests.log
[ $rc -eq 0 ] || exit 1
python3 profiles/ab_inprocess.py --batch-log2 27 --steps 12 --rounds 5 f=0 f=0x80000 > gpurun_out/r04/ab_piped.log 2>&1
tail -3 gpurun_out/r04/ab_piped.log
python3 profiles/ab_inprocess.py --batch-log2 27 --steps 12 --rounds 3 --one-stream f=0 f=0x80000 > gpurun_out/r04/ab_piped_one.log 2>&1
tail -3 gpurun_out/r04/ab_piped_one.log
