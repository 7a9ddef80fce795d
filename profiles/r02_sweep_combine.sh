# threads per workgroup of the combine kernel (GF2_OPT_COMBINE_THREADS = option 5) on the default bench step, same box, alternating
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 %.4g %.4f ms frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
for i in 1 2; do
  run "--opt 5=1024" threads=1024
  run "--opt 5=256" threads=256
  run "--opt 5=128" threads=128
  run "--opt 5=64" threads=64
done
run "--opt 5=1024 --one-stream" one-stream-1024
run "--opt 5=256 --one-stream" one-stream-256
