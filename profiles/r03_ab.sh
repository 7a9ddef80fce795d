# Round 3, one box.  (1) per-kernel durations (rocprofv3 --kernel-trace over profiles/time_slabs.py, 2^21 samples per call, one stream)
# of the working tree's library and of the variants in scratch_ab/ (profiles/build_variant.sh): `old` = gf2_slabs.hip before the
# records' header moved and the gather kernel took slot pairs from four tiles; the others are timing experiments with parts of the
# gather kernel taken out (wrong results: time_slabs.py does not check any).  (2) the default two-stream bench, new against old,
# interleaved.  (3) the streamed RREF with and without look-ahead.  (4) wall times of the drop-in calls.
root=$(pwd); out=$root/gpurun_out/r03; mkdir -p $out
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
use() { if [ $1 = new ]; then cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so; else cp scratch_ab/$1.so quantum_css_codes_amd/libgf2hip.so; fi; }
for which in new old ${VARIANTS:-nolookup seqident noident norec notake nomem}; do
  use $which
  (cd /tmp && TMPDIR=/tmp SLAB_LOG2_BATCH=21 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ab_$which -- python3 $root/profiles/time_slabs.py > $out/ab_$which.txt 2> $out/ab_$which.err) || { echo "$which failed"; tail -3 $out/ab_$which.err; continue; }
  python3 profiles/summarize.py $(find $out/ab_$which -name '*kernel_trace.csv') > $out/ab_$which.md
  echo "== $which: $(cat $out/ab_$which.txt)"; grep -E "slab_(gather|compact|combine|redo)" $out/ab_$which.md | awk -F'|' '$3+0 > 100 {print $2, $3, "calls", $8, "mean us", $9}'
done
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 %.4g /s %.4f ms frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
for i in 1 2; do
  use new; run "" new
  use old; run "" old
done
use new; run "--one-stream" new-1s
use old; run "--one-stream" old-1s
use new
echo "== rref 32768 x 65536, look-ahead on / off"
python3 profiles/time_rref.py 32768 65536 1
GF2_FLAGS=0x10000 python3 profiles/time_rref.py 32768 65536 1
python3 profiles/time_rref.py 16384 32768 1
GF2_FLAGS=0x10000 python3 profiles/time_rref.py 16384 32768 1
echo "== api wall times (threads: default, then 1)"
python3 profiles/time_api.py
GF2_HOST_THREADS=1 python3 profiles/time_api.py
