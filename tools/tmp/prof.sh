cd /root/repo
bash tools/collect_profiles.sh r05b > gpurun_out/prof_r05b.log 2>&1
python tools/profile_layers.py 1024 6 > gpurun_out/prof_r05b/per_layer_times_1024.txt 2>&1
python tools/profile_layers.py 724 6 > gpurun_out/prof_r05b/per_layer_times_724.txt 2>&1
python bench.py > gpurun_out/prof_r05b/bench_line.json 2> gpurun_out/prof_r05b/bench_line.err
STX_GRAM=bf3 STX_SYMM=bf3 STX_SUMS_LATE=0 python bench.py --no-cpu-baseline --no-wall-clock > gpurun_out/prof_r05b/bench_line_start_of_session.json 2>/dev/null
python bench.py --no-cpu-baseline --no-wall-clock > gpurun_out/prof_r05b/bench_line_2.json 2>/dev/null
python tools/scale_steps.py > gpurun_out/prof_r05b/scale_steps.txt 2>&1
bash tools/time_cli.sh all > gpurun_out/prof_r05b/time_cli.log 2>&1
python tools/lbfgs_step.py > gpurun_out/prof_r05b/lbfgs_step.txt 2>&1
tail -3 gpurun_out/prof_r05b/bench_line.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prof_r05b/bench_line*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],1), d.get('wall_clock_s'), round(d['roofline']['frac'],3))
    except Exception as ex: print(f, 'ERR', ex)
PY
cat gpurun_out/prof_r05b/time_cli.log | grep wall
