# Same-box A/B of the default two-stream step under context options (bench.py --opt K=V), alternating, N rounds; then the
# per-kernel durations of a one-stream run for each setting.   bash profiles/r03_ab_opt.sh "8=0 8=1" 4
root=$(pwd); out=$root/gpurun_out/r03; mkdir -p $out
: > $out/ab_opt.txt
for i in $(seq 1 ${2:-4}); do
  for o in $1; do
    python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --opt $o 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o %.4f' % d['roofline']['frac'])" | tee -a $out/ab_opt.txt
  done
done
python3 - <<PY
import collections, statistics
runs = collections.defaultdict(list)
for line in open("$out/ab_opt.txt"):
    k, v = line.split(); runs[k].append(float(v))
for k, v in runs.items():
    print(k, "median %.4f  min %.4f  max %.4f  n %d" % (statistics.median(v), min(v), max(v), len(v)))
PY
cd /tmp && export TMPDIR=/tmp
for o in $1; do
  rm -rf /tmp/tr_$o
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$o -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24 --one-stream --opt $o > /dev/null 2>&1
  echo "one stream, --opt $o:"; python3 $root/profiles/summarize.py $(find /tmp/tr_$o -name '*kernel_trace.csv') | grep -E "slab_(gather|compact|combine)" | cut -c1-150
done
