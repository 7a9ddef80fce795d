# Round 5, VERDICT r04 item 2: cache policy of the once-read streams of the slab pipeline.  Four libraries (profiles/build_variant.sh
# nt<k> - -DGF2_SLAB_NT=<k>: 1 = compact's row loads nt, 2 = gather's identity loads nt, 3 = both; the working tree's = default
# policy) alternate on ONE box, ROUNDS rounds of the default two-stream bench each, then the pass size 2^20 .. 2^22 for each.
#   bash profiles/r05_nt.sh [rounds]      -> gpurun_out/r05/nt_ab.txt, nt_pass.txt
root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
cp quantum_css_codes_amd/libgf2hip.so /tmp/base.so
use() { if [ $1 = base ]; then cp /tmp/base.so $root/quantum_css_codes_amd/libgf2hip.so; else cp $root/scratch_ab/$1.so $root/quantum_css_codes_amd/libgf2hip.so; fi; }
frac() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 %.4f' % d['roofline']['frac'])"; }
: > $out/nt_ab.txt
for i in $(seq 1 ${1:-6}); do
  for w in base nt1 nt2 nt3; do
    use $w
    python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | frac $w | tee -a $out/nt_ab.txt
  done
done
: > $out/nt_pass.txt
for k in 20 21 22; do
  for w in base nt1 nt3; do
    use $w
    for i in 1 2; do
      python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --slab-pass-log2 $k 2>/dev/null | frac ${w}_pass$k | tee -a $out/nt_pass.txt
    done
  done
done
use base
python3 - <<PY
import collections, statistics
for f in ("$out/nt_ab.txt", "$out/nt_pass.txt"):
    runs = collections.defaultdict(list)
    for line in open(f):
        k, v = line.split(); runs[k].append(float(v))
    for k, v in runs.items():
        print(k, "median %.4f  min %.4f  max %.4f  n %d" % (statistics.median(v), min(v), max(v), len(v)))
PY
