#!/bin/bash
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${1:-r06_phase}
mkdir -p "$OUT"; cd "$R"
run() { local name=$1; shift; echo "## $name: $*"; env "$@" 2>&1 | grep -v "^fp16 MFMA\|^split of"; }
{
for rep in 1 2 3; do
for spec in "X=1|256 256 256 256 2 0" "X=1|512 512 128 128 2 0" "X=1|128 128 512 512 2 0" "X=1|256 256 256 256 2 1" "INJECT=1|256 256 256 256 2 1" "X=1|64 64 1024 1024 3 0" "PIN=1 INJECT=1|64 64 1024 1024 3 1" "X=1|64 128 512 512 2 0" "X=1|128 64 512 512 3 1" "PIN=1|256 256 256 256 2 1" "X=1|256 256 181 181 2 0" "X=1|512 512 64 64 2 0"; do
  envs=${spec%%|*}; args=${spec##*|}
  run base $envs tools/ubench/bin/h2conv_bench_timing $args
  run early $envs tools/ubench/bin/h2conv_bench_phase $args
done; done
} > "$OUT/ab.txt" 2>&1
grep -c " ms " "$OUT/ab.txt"
