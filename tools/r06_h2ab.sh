#!/bin/bash
# A/B of the conv_h2 harness builds on one box: tools/ubench/bin/h2conv_bench_{base,timing}
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-r06_h2ab}
mkdir -p "$OUT"
cd "$R"
run() {  # name, env..., args
  local name=$1; shift
  echo "## $name: $*"
  env "$@" 2>&1 | grep -v "^fp16 MFMA\|^split of"
}
shapes=("64 64 1024 1024 3 0" "64 64 1024 1024 3 1" "64 128 512 512 2 0" "128 64 512 512 3 1" "128 128 512 512 2 0" "128 128 512 512 2 1" "256 256 256 256 2 0" "256 256 256 256 2 1" "512 512 128 128 2 0" "512 512 64 64 2 0" "256 256 181 181 2 0" "512 512 91 91 2 0")
{
for rep in 1 2; do
for s in "${shapes[@]}"; do
  run base tools/ubench/bin/h2conv_bench_base $s
  run "new, one item per workgroup" STX_H2_PERSIST=0 tools/ubench/bin/h2conv_bench_timing $s
  run "new, persistent" tools/ubench/bin/h2conv_bench_timing $s
done
done
run base PIN=1 INJECT=1 tools/ubench/bin/h2conv_bench_base 64 64 1024 1024 3 1
run new0 STX_H2_PERSIST=0 PIN=1 INJECT=1 tools/ubench/bin/h2conv_bench_timing 64 64 1024 1024 3 1
run new PIN=1 INJECT=1 tools/ubench/bin/h2conv_bench_timing 64 64 1024 1024 3 1
run base PIN=1 INJECT=1 tools/ubench/bin/h2conv_bench_base 128 128 512 512 2 1
run new0 STX_H2_PERSIST=0 PIN=1 INJECT=1 tools/ubench/bin/h2conv_bench_timing 128 128 512 512 2 1
run new PIN=1 INJECT=1 tools/ubench/bin/h2conv_bench_timing 128 128 512 512 2 1
run base INJECT=1 tools/ubench/bin/h2conv_bench_base 256 256 256 256 2 1
run new INJECT=1 tools/ubench/bin/h2conv_bench_timing 256 256 256 256 2 1
} > "$OUT/h2conv_ab.txt" 2>&1
grep -c "ms" "$OUT/h2conv_ab.txt"
