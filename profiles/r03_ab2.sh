root=$(pwd); out=$root/gpurun_out/r03; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slab or sparse or monte_carlo or config4 or two_contexts" > $out/t_slab2.log 2>&1; tail -n 3 $out/t_slab2.log
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
use() { if [ $1 = new ]; then cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so; else cp scratch_ab/$1.so quantum_css_codes_amd/libgf2hip.so; fi; }
for which in new prev; do
  use $which
  (cd /tmp && TMPDIR=/tmp SLAB_LOG2_BATCH=21 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ab2_$which -- python3 $root/profiles/time_slabs.py > $out/ab2_$which.txt 2> $out/ab2_$which.err) || { echo "$which failed"; tail -3 $out/ab2_$which.err; continue; }
  python3 profiles/summarize.py $(find $out/ab2_$which -name '*kernel_trace.csv') > $out/ab2_$which.md
  echo "== $which: $(cat $out/ab2_$which.txt)"; grep -E "slab_(gather|compact)" $out/ab2_$which.md | awk -F'|' '$3+0 > 100 {print $2, $3, "calls", $8, "mean us", $9}'
done
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --batch-log2 24 $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 %.4g /s %.4f ms frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
for i in 1 2 3; do
  use new; run "" new
  use prev; run "" prev
done
use new; run "--one-stream" new-1s
use prev; run "--one-stream" prev-1s
use new
