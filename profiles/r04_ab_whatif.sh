# Same-box timing of the working tree's library against a WHAT-IF variant (scratch_ab/<name>.so: wrong results, python3 -O skips the
# checks of bench.py): per-kernel durations of a one-stream run each, then the default two-stream step, alternating.   bash profiles/r04_ab_whatif.sh <name> [rounds]
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
use() { if [ $1 = new ]; then cp /tmp/new.so $root/quantum_css_codes_amd/libgf2hip.so; else cp $root/scratch_ab/$1.so $root/quantum_css_codes_amd/libgf2hip.so; fi; }
cd /tmp && export TMPDIR=/tmp
for w in new $1; do
  use $w; rm -rf /tmp/tr_$w
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$w -- python3 -O $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24 --one-stream > /dev/null 2>&1
  echo "one stream, $w:"; python3 $root/profiles/summarize.py $(find /tmp/tr_$w -name '*kernel_trace.csv') | grep -E "slab_(gather|compact|combine)" | head -3 | cut -c1-150
done
cd $root
: > $out/ab_whatif.txt
for i in $(seq 1 ${2:-4}); do
  for w in new $1; do
    use $w
    python3 -O bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w %.4f' % d['roofline']['frac'])" | tee -a $out/ab_whatif.txt
  done
done
use new
python3 - <<PY
import collections, statistics
runs = collections.defaultdict(list)
for line in open("$out/ab_whatif.txt"):
    k, v = line.split(); runs[k].append(float(v))
for k, v in runs.items():
    print(k, "median %.4f  min %.4f  max %.4f  n %d" % (statistics.median(v), min(v), max(v), len(v)))
PY
