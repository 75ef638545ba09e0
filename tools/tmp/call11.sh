cd /root/repo
timeout 900 python -m pytest tests/test_gpu_tile_path.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -3
for r in 1 2; do for n in 2 3 4 1; do
  STX_STREAMS_PER_GPU=$n python bench.py --no-cpu-baseline --no-wall-clock --steady-seconds 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $n', round(d['value'],1), round(d['steady']['value'],1))"
done; done
