#!/bin/bash
# Per-dispatch fetch / write of one 1024^2 tile evaluation:  bash tools/collect_layers.sh <tag>
set -u
TAG=${1:-x}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/layers_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TILE="python $R/tools/bench_tile.py ${2:-1024} 2"
find_csv() { find "$1" -name "*_$2.csv" | head -1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $TILE > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $TILE > "$OUT/write.log" 2>&1
python "$R/tools/pmc_layers.py" "$(find_csv "$OUT/fetch" counter_collection)" "$(find_csv "$OUT/write" counter_collection)" > "$OUT/layers.txt" 2> "$OUT/err.log"
rm -rf "$OUT/fetch" "$OUT/write"
cat "$OUT/layers.txt" "$OUT/err.log"
