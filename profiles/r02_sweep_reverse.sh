set -e
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "slab or sparse" > gpurun_out/r02/slab_tests.log 2>&1 || (tail -30 gpurun_out/r02/slab_tests.log; exit 1)
for rev in 0 1 0 1; do
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary --opt 2=$rev >> gpurun_out/r02/rev_$rev.json 2>> gpurun_out/r02/rev.err
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary --one-stream --opt 2=$rev >> gpurun_out/r02/rev1s_$rev.json 2>> gpurun_out/r02/rev.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02/rev*.json')):
    for line in open(f).read().strip().splitlines():
        d=json.loads(line); print(f, '%.4g'%d['value'], '%.4f'%d['ms_per_step'], '%.4f'%d['roofline']['frac'])
PY
