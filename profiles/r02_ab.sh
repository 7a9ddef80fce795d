# A/B on one box: the working tree's library against the library of HEAD (built into /tmp)
set -e
mkdir -p gpurun_out/r02
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 %.4g %.4f ms frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slab or sparse or two_contexts or monte_carlo_n4096" 2>&1 | tail -3
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
for i in 1 2; do
  cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so; run "" new
  cp scratch_ab/old.so quantum_css_codes_amd/libgf2hip.so; run "" old
done
cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so; run "--one-stream" new-1s
cp scratch_ab/old.so quantum_css_codes_amd/libgf2hip.so; run "--one-stream" old-1s
cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so
