#!/bin/bash
# Round 6: the rocprofv3 evidence, per-layer times, per-scale steps, whole command-line runs (one box, one call).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-r06_prof}
mkdir -p "$OUT"
cd "$R"
python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
python tools/profile_layers.py 1024 5 > "$OUT/per_layer_times.txt" 2>&1
python tools/profile_layers.py 724 5 >> "$OUT/per_layer_times.txt" 2>&1
python tools/scale_steps.py > "$OUT/scale_steps.txt" 2>&1
bash tools/time_cli.sh all > "$OUT/time_cli.log" 2>&1
bash tools/collect_profiles.sh ${2:-r06} > "$OUT/collect.log" 2>&1
cp gpurun_out/prof_${2:-r06}/*.txt gpurun_out/prof_${2:-r06}/*.json "$OUT/" 2>/dev/null
python bench.py --no-cpu-baseline > "$OUT/bench_line_2.json" 2>> "$OUT/bench.err"
ls -la "$OUT"
