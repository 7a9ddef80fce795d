#!/bin/bash
mkdir -p gpurun_out/r04
python3 profiles/r04_lead.py > gpurun_out/r04/lead.log 2>&1
cat gpurun_out/r04/lead.log
