root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $out/full_tests.log 2>&1; echo "tests rc=$?" >> $out/full_tests.log
tail -4 $out/full_tests.log
python3 bench.py --steps 20 --warmup 3 > $out/full_bench.json 2> $out/full_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.loads(open("$out/full_bench.json").read().strip().splitlines()[-1])
print("frac", d["roofline"]["frac"], "ms", d["ms_per_step"])
s = d["secondary"]
print("rw1536", s["read_write_1536B"]["value"], s["read_write_1536B"]["roofline"]["frac"])
print("one_stream", s["one_stream"]["roofline"]["frac"], "mc", s["monte_carlo_end_to_end"]["value"])
print("literal", s.get("configs4_literal_1e8"))
print({k: (v["ms"], v["GB/s"]) for k, v in s["rref"].items()})
PY
