# Same-box A/B of routing flags on the default two-stream bench line: bash profiles/r04_ab_flags.sh "<flags A> <flags B> ..." [rounds]
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
: > $out/ab_flags.txt
for i in $(seq 1 ${2:-4}); do
  for f in $1; do
    python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --ctx-flags $f 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f %.4f %.3f' % (d['roofline']['frac'], d['ms_per_step']))" | tee -a $out/ab_flags.txt
  done
done
python3 - <<PY
import collections, statistics
runs = collections.defaultdict(list)
for line in open("$out/ab_flags.txt"):
    k, v, _ = line.split(); runs[k].append(float(v))
for k, v in runs.items():
    print(k, "median %.4f  min %.4f  max %.4f  n %d" % (statistics.median(v), min(v), max(v), len(v)))
PY
