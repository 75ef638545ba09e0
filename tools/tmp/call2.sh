cd /root/repo
mkdir -p gpurun_out/c2
timeout 900 python -m pytest tests/test_gpu_tile_path.py tests/test_gpu_determinism.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -4
for i in 1 2; do
  STX_SUMS_LATE=0 python bench.py --no-cpu-baseline --no-wall-clock --steady-seconds 3 > gpurun_out/c2/bench_early_$i.json 2>gpurun_out/c2/bench_early_$i.err
  python bench.py --no-cpu-baseline --no-wall-clock --steady-seconds 3 > gpurun_out/c2/bench_late_$i.json 2>gpurun_out/c2/bench_late_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c2/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],1), round(d['steady']['value'],1))
    except Exception as ex: print(f, 'ERR', ex)
PY
python tools/scale_steps.py > gpurun_out/c2/scale_steps.txt 2>&1; tail -12 gpurun_out/c2/scale_steps.txt
STX_SUMS_LATE=0 python tools/scale_steps.py > gpurun_out/c2/scale_steps_early.txt 2>&1; tail -12 gpurun_out/c2/scale_steps_early.txt
