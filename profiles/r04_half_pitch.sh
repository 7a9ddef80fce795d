#!/bin/bash
mkdir -p gpurun_out/r04
cp quantum_css_codes_amd/libgf2hip.so /tmp/keep.so; cp scratch_ab/halfpitch.so quantum_css_codes_amd/libgf2hip.so
python3 -O profiles/r04_half_pitch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/half_pitch.log
cp /tmp/keep.so quantum_css_codes_amd/libgf2hip.so
cat gpurun_out/r04/half_pitch.log
