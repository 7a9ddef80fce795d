# Round-4 baseline on this round's box: default two-stream line, one-stream line, per-kernel split of the one-stream run.
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
python3 bench.py --steps 20 --warmup 3 > $out/base_default.json 2> $out/base_default.err
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --one-stream > $out/base_one.json 2> $out/base_one.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_base -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-settle --batch-log2 24 --one-stream > /dev/null 2>&1
python3 $root/profiles/summarize.py $(find /tmp/tr_base -name '*kernel_trace.csv') > $out/base_one_trace.md
cd $root
head -8 $out/base_one_trace.md | cut -c1-160
python3 - <<PY
import json
for f in ("base_default", "base_one"):
    d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["roofline"]["frac"], d["ms_per_step"])
PY
