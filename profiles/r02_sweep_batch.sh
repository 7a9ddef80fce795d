mkdir -p gpurun_out/r02
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
for cfg in "--batch-log2 24" "--batch-log2 25" "--batch-log2 24 --slab-pass-log2 20" "--batch-log2 22" "--batch-log2 24 --opt 2=0" "--batch-log2 24 --one-stream"; do
  echo "$cfg" >> gpurun_out/r02/sweep_batch.txt
  timeout -k 10 200 $B $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g %.4f ms frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))" >> gpurun_out/r02/sweep_batch.txt
done
cat gpurun_out/r02/sweep_batch.txt
