# Same-box A/B of bench options on the default two-stream line: bash profiles/r04_ab_opts.sh "<args A>|<args B>|..." [rounds]
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
: > $out/ab_opts.txt
IFS='|' read -ra variants <<< "$1"
for i in $(seq 1 ${2:-3}); do
  for idx in "${!variants[@]}"; do
    python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary ${variants[$idx]} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('v$idx %.4f %.3f' % (d['roofline']['frac'], d['ms_per_step']))" | tee -a $out/ab_opts.txt
  done
done
python3 - <<PY
import collections, statistics
runs = collections.defaultdict(list)
for line in open("$out/ab_opts.txt"):
    k, v, _ = line.split(); runs[k].append(float(v))
names = """$1""".split("|")
for k, v in sorted(runs.items()):
    print(k, "[%s]" % names[int(k[1:])], "median %.4f  min %.4f  max %.4f  n %d" % (statistics.median(v), min(v), max(v), len(v)))
PY
