B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
for cfg in "" "--opt 1=64" "--opt 1=32" "--opt 1=96" "" "--opt 1=64"; do
  echo -n "[$cfg] "
  timeout -k 10 200 $B $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g %.4f ms frac %.4f of measured %.3f (%.0f GB/s)' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_of_measured_peak'], d['roofline']['measured_read_peak_GBs']))"
done
