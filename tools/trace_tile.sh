#!/bin/bash
# Per-dispatch durations of one tile evaluation:  bash tools/trace_tile.sh <tag> [size] [filter-regex]
# (rocprofv3 --kernel-trace over tools/bench_tile.py)
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-x}; SIZE=${2:-1024}; PAT=${3:-.}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$TAG -- python "$R/tools/bench_tile.py" "$SIZE" 4 > "$OUT/trace_$SIZE.log" 2>&1
python "$R/tools/trace_layers.py" "$(find /tmp/tr_$TAG -name '*kernel_trace.csv' | head -1)" > "$OUT/trace_$SIZE.txt" 2>&1
grep -v "^[EW]2026" "$OUT/trace_$SIZE.log" | tail -1
grep -E "$PAT|^kernels" "$OUT/trace_$SIZE.txt"
