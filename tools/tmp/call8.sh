cd /root/repo
timeout 900 python -m pytest tests/test_gpu_tile_path.py -q -x 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "543 or 965" 2>&1 | tail -4
python tools/profile_layers.py 965 6 2>&1 | grep -E "pool|TOTAL"
for i in 1 2; do
python bench.py --no-cpu-baseline --no-wall-clock --steady-seconds 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value'],1), round(d['steady']['value'],1))"
done
