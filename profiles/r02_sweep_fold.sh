# the combine step inside the next pass' compact kernel (GF2_F_COMBINE_FOLDED = 0x8000) against a combine kernel after every pass
# (default) on the default bench step, same box, alternating
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 %.4g %.4f ms frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
for i in 1 2 3; do
  run "--ctx-flags 0x8000" folded
  run "" unfolded
done
run "--one-stream --ctx-flags 0x8000" one-stream-folded
run "--one-stream" one-stream-unfolded
