# Round 4: stored syndromes through the slab pipeline -- parity tests, then the bench line with secondary.read_write_1536B
root=$(pwd); out=$root/gpurun_out/r04; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slab or config4_code_itself or redo" > $out/syn_tests.log 2>&1; echo "tests rc=$?" >> $out/syn_tests.log
tail -5 $out/syn_tests.log
timeout -k 10 600 python3 -m pytest tests/test_rccl.py -x -q -m gpu > $out/rccl_tests.log 2>&1; echo "rccl rc=$?" >> $out/rccl_tests.log
tail -5 $out/rccl_tests.log
python3 bench.py --steps 20 --warmup 3 > $out/syn_bench.json 2> $out/syn_bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.loads(open("$out/syn_bench.json").read().strip().splitlines()[-1])
print("frac", d["roofline"]["frac"], "ms", d["ms_per_step"])
s = d["secondary"]
print("rw1536", s["read_write_1536B"]["value"], s["read_write_1536B"]["roofline"]["frac"])
print("one_stream", s["one_stream"]["roofline"]["frac"], "mc", s["monte_carlo_end_to_end"]["value"])
print("literal", s.get("configs4_literal_1e8"))
print({k: (v["ms"], v["GB/s"]) for k, v in s["rref"].items()})
PY
