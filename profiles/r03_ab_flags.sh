# Same-box A/B of the default two-stream step under different GF2_FLAGS values, alternating, N rounds:
#   bash profiles/r03_ab_flags.sh "0 0x40000" 4 [bench args]
root=$(pwd); out=$root/gpurun_out/r03; mkdir -p $out
: > $out/ab_flags.txt
for i in $(seq 1 ${2:-4}); do
  for f in $1; do
    GF2_FLAGS=$f python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary ${@:3} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f %.4f' % d['roofline']['frac'])" | tee -a $out/ab_flags.txt
  done
done
python3 - <<PY
import collections, statistics
runs = collections.defaultdict(list)
for line in open("$out/ab_flags.txt"):
    k, v = line.split(); runs[k].append(float(v))
for k, v in runs.items():
    print(k, "median %.4f  min %.4f  max %.4f  n %d" % (statistics.median(v), min(v), max(v), len(v)))
PY
