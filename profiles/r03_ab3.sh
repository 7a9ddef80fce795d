# Same-box A/B of the default two-stream step: the working tree's library against scratch_ab/<name>.so, alternating, N rounds; prints
# every run and the medians.   bash profiles/r03_ab3.sh "<variants>" <rounds>
root=$(pwd); out=$root/gpurun_out/r03; mkdir -p $out
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
use() { if [ $1 = new ]; then cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so; else cp scratch_ab/$1.so quantum_css_codes_amd/libgf2hip.so; fi; }
: > $out/ab3.txt
for i in $(seq 1 ${2:-5}); do
  for which in new $1; do
    use $which
    python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --batch-log2 24 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which %.4f' % d['roofline']['frac'])" | tee -a $out/ab3.txt
  done
done
use new
python3 - <<PY
import collections, statistics
runs = collections.defaultdict(list)
for line in open("$out/ab3.txt"):
    k, v = line.split(); runs[k].append(float(v))
for k, v in runs.items():
    print(k, "median %.4f  min %.4f  max %.4f  n %d" % (statistics.median(v), min(v), max(v), len(v)))
PY
