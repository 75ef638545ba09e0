#!/bin/bash
# Round 6, second GPU call: the changed tests (precision A/B, e2e legs, the stable fixture), then the whole suite.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r06_call2
mkdir -p "$OUT/e2e"
cd "$R"
export STX_PARITY_STATS=$OUT/tile_parity_stats.txt STX_E2E_DUMP=$OUT/e2e STX_PRECISION_STATS=$OUT/precision_ab.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -k "no_worse or fp16_split" > "$OUT/pytest_precision.log" 2>&1
echo "precision rc $?"; tail -3 "$OUT/pytest_precision.log"
timeout 1500 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_cli.py -q -s -k "lbfgs or config4" > "$OUT/pytest_e2e.log" 2>&1
echo "e2e rc $?"; tail -3 "$OUT/pytest_e2e.log"
timeout 2400 python -m pytest tests -m gpu -q -x > "$OUT/pytest_all.log" 2>&1
echo "all rc $?"; tail -3 "$OUT/pytest_all.log"
