set -e
mkdir -p gpurun_out/r02
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02/smoke0.log 2>&1
for p in 21 20 19 18 17 16; do
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary --slab-pass-log2 $p > gpurun_out/r02/pass_$p.json 2> gpurun_out/r02/pass_$p.err
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary --one-stream --slab-pass-log2 $p > gpurun_out/r02/pass1s_$p.json 2>> gpurun_out/r02/pass_$p.err
done
for p in 20 18; do
  python bench.py --steps 100 --warmup 10 --batch-log2 22 --no-cpu-baseline --no-secondary --slab-pass-log2 $p > gpurun_out/r02/pass_b22_$p.json 2>> gpurun_out/r02/pass_$p.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/pass*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
