#!/bin/bash
# per-workgroup histogram rows for the riding combine step (default) against global atomics (GF2_F_COMBINE_ATOMICS = 0x80000)
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "slab or mc or monte or records or three_routes" > gpurun_out/r04/part_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r04/part_tests.log
tail -3 gpurun_out/r04/part_tests.log
python3 profiles/ab_inprocess.py --batch-log2 27 --steps 12 --rounds 6 f=0 f=0x80000 > gpurun_out/r04/ab_part.log 2>&1
tail -3 gpurun_out/r04/ab_part.log
