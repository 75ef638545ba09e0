cd /root/repo
mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/final/pytest_tail.txt
cat gpurun_out/final/pytest_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench_line.err
python tools/scale_steps.py > gpurun_out/final/scale_steps.txt 2>&1
bash tools/time_cli.sh all > gpurun_out/final/time_cli.log 2>&1
python tools/profile_layers.py 965 6 > gpurun_out/final/per_layer_965.txt 2>&1
python tools/profile_layers.py 724 6 > gpurun_out/final/per_layer_724.txt 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench_line.json').read().strip().splitlines()[-1]); print('bench', round(d['value'],1), d.get('wall_clock_s'), round(d['roofline']['frac'],3), d['roofline']['traffic'])
PY
grep wall gpurun_out/final/time_cli.log; cat gpurun_out/final/scale_steps.txt; grep TOTAL gpurun_out/final/per_layer_*.txt
