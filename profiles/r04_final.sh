# Round 4, final state: smoke, the whole GPU suite, the bench line (kept as profiles/r04_bench_default.json), five further default runs
# interleaved with --one-stream runs (the spread of one box), the in-process A/B of the round's one kept switch
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/final_smoke.log
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $out/final_tests.log 2>&1; echo "tests rc=$?" >> $out/final_tests.log; tail -3 $out/final_tests.log
python3 bench.py --steps 20 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
: > $out/final_runs.txt
for i in 1 2 3 4 5; do
  for v in "" "--one-stream"; do
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $v 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-two-streams} %.4f %.3f ms' % (d['roofline']['frac'], d['ms_per_step']))" | tee -a $out/final_runs.txt
  done
done
python3 profiles/ab_inprocess.py --rounds 6 --steps 10 f=0 f=0x40000 > $out/final_ab.log 2>&1; tail -3 $out/final_ab.log
python3 - <<PY
import json
d = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("frac", d["roofline"]["frac"], "ms", d["ms_per_step"], "traffic/alg", d["roofline"]["traffic"] / d["roofline"]["algorithmic_bytes_per_launch"])
s = d["secondary"]
print("rw1536", s["read_write_1536B"]["roofline"]["frac"], "one_stream", s["one_stream"]["roofline"]["frac"], "mc", s["monte_carlo_end_to_end"]["value"])
print("literal", s["configs4_literal_1e8"]["ms"])
print({k: (round(v["ms"], 3), round(v["GB/s"])) for k, v in s["rref"].items()})
PY
