#!/bin/bash
# Round 6: the whole GPU suite on the current library, a bench line, where the whole run's time outside the steps goes.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-r06_call3}
mkdir -p "$OUT/e2e"
cd "$R"
export STX_PARITY_STATS=$OUT/tile_parity_stats.txt STX_E2E_DUMP=$OUT/e2e STX_PRECISION_STATS=$OUT/precision_ab.txt
timeout 3000 python -m pytest tests -m gpu -q > "$OUT/pytest_all.log" 2>&1
echo "all rc $?"; tail -5 "$OUT/pytest_all.log"
unset STX_PARITY_STATS STX_E2E_DUMP STX_PRECISION_STATS
python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
python -c "
import json,sys
d=json.load(open('$OUT/bench_line.json'))
print('bench', d['value'], d['ms_per_step'], 'wall', d.get('wall_clock_s'), 'fp32', d.get('fp32_kernels',{}).get('value'))
print('dominant', d['roofline'].get('dominant'))
"
python tools/profile_cli.py > "$OUT/profile_cli.txt" 2>&1
head -40 "$OUT/profile_cli.txt"
