cd /root/repo
timeout 900 python -m pytest tests/test_gpu_tile_path.py -q -x 2>&1 | tail -2
for t in 0 1 0 1; do
  echo "== STX_H2_TOUCH=$t"
  STX_H2_TOUCH=$t python tools/profile_layers.py 1024 6 2>&1 | grep -E "bwd conv1_2|bwd conv2_2|bwd conv3_2|bwd conv4_2|bwd conv4_3|bwd conv3_3|TOTAL"
done
for i in 1 2; do for t in 0 1; do
  STX_H2_TOUCH=$t python bench.py --no-cpu-baseline --no-wall-clock --steady-seconds 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('touch $t', round(d['value'],1), round(d['steady']['value'],1))"
done; done
