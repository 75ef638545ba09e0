#!/bin/bash
# Collects the rocprofv3 evidence for one build state on the GPU box (run through gpurun from the
# repository root):   bash tools/collect_profiles.sh r02
# Writes summaries to gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/.
#   1. single-stream kernel trace + stats of 1024^2 tile evaluations (2 warm-up + 7: per-kernel
#      durations; the last column is per tile evaluation)
#   2. SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE per kernel (matrix-pipe utilisation)
#   3. FETCH_SIZE and WRITE_SIZE passes over bench.py (HBM traffic per tile-iteration)
# Counters are collected in runs of their own, with --kernel-trace only (MI355X_MICROARCH.md).
set -u
TAG=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-wall-clock --steady-seconds 0 --no-kernel-records --no-fp32-leg"
TILE="python $R/tools/bench_tile.py 1024 7"
find_csv() { find "$1" -name "*_$2.csv" | head -1; }

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $TILE > "$OUT/stats.log" 2>&1
python "$R/tools/kernel_stats.py" "$(find_csv "$OUT/stats" kernel_trace)" 9 > "$OUT/single_tile_1024_kernel_stats.txt" 2>> "$OUT/stats.log"

rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" -- $TILE > "$OUT/mfma.log" 2>&1
python "$R/tools/pmc_mfma.py" "$(find_csv "$OUT/mfma" counter_collection)" "$(find_csv "$OUT/mfma" kernel_trace)" > "$OUT/mfma_utilisation_pmc.txt" 2>> "$OUT/mfma.log"

rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.log" 2>&1
python "$R/tools/pmc_traffic.py" "$(find_csv "$OUT/fetch" counter_collection)" "$(find_csv "$OUT/write" counter_collection)" 2048 4 > "$OUT/hbm_traffic_pmc.json" 2>> "$OUT/write.log"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -- $BENCH > "$OUT/bench.log" 2>&1
python "$R/tools/kernel_stats.py" "$(find_csv "$OUT/bench" kernel_trace)" 16 > "$OUT/bench_2048_kernel_stats.txt" 2>> "$OUT/bench.log"
rm -rf "$OUT/stats" "$OUT/mfma" "$OUT/fetch" "$OUT/write" "$OUT/bench"
ls -la "$OUT"
