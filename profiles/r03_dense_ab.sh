cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
for i in 1 2 3; do for w in new old; do
 if [ $w = new ]; then cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so; else cp scratch_ab/dense_old.so quantum_css_codes_amd/libgf2hip.so; fi
 python3 bench.py --algo dense --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --batch-log2 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '%.4g' % d['value'], d['ms_per_step'])"
done; done
cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so
