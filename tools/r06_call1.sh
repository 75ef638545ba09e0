#!/bin/bash
# Round 6, first GPU call: cycle budget of the shallow layers (baseline for the kernel work), the new
# precision / A-B tests, the e2e fixtures' GPU outcomes for the offline comparison, a bench line.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r06_call1
mkdir -p "$OUT/e2e"
cd "$R"
B=tools/ubench/bin/h2conv_bench_timing
{
  $B 64 64 1024 1024 3 0
  $B 64 64 1024 1024 1 0
  PIN=1 INJECT=1 $B 64 64 1024 1024 3 1
  PIN=1 $B 64 64 1024 1024 3 1
  $B 64 128 512 512 2 0
  $B 128 64 512 512 3 1
  $B 128 128 512 512 2 0
  PIN=1 INJECT=1 $B 128 128 512 512 2 1
  $B 256 256 256 256 2 0
  $B 512 512 128 128 2 0
} > "$OUT/h2conv_timing.txt" 2>&1
export STX_PARITY_STATS=$OUT/tile_parity_stats.txt STX_E2E_DUMP=$OUT/e2e
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -k "no_worse" -s > "$OUT/pytest_precision.log" 2>&1
echo "precision rc $?"
timeout 1500 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_cli.py -q -s -k "lbfgs_avgpool or config4" > "$OUT/pytest_e2e.log" 2>&1
echo "e2e rc $?"
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -s -k "sc_grad_tile" > "$OUT/pytest_fullsize.log" 2>&1
echo "fullsize rc $?"
python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench_line.json"
