cd /root/repo
mkdir -p gpurun_out/c1
timeout 120 ./tools/ubench/bin/symm_bench > gpurun_out/c1/symm_ab.txt 2>&1
timeout 120 ./tools/ubench/bin/gram_bench > gpurun_out/c1/gram_ab.txt 2>&1
cat gpurun_out/c1/symm_ab.txt gpurun_out/c1/gram_ab.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gram or style_terms" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "gram or sc_grad_tile_at" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_tile_path.py -q -x 2>&1 | tail -5
for i in 1 2; do
  STX_GRAM=bf3 STX_SYMM=bf3 python bench.py --no-cpu-baseline --no-wall-clock --steady-seconds 3 > gpurun_out/c1/bench_bf3_$i.json 2>gpurun_out/c1/bench_bf3_$i.err
  python bench.py --no-cpu-baseline --no-wall-clock --steady-seconds 3 > gpurun_out/c1/bench_h2_$i.json 2>gpurun_out/c1/bench_h2_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c1/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],1), round(d['steady']['value'],1))
    except Exception as ex: print(f, 'ERR', ex)
PY
python tools/profile_layers.py 1024 6 > gpurun_out/c1/per_layer.txt 2>&1; grep -i "gram\|symm\|TOTAL" gpurun_out/c1/per_layer.txt
