#!/bin/bash
# Register table (and listing) of five conv_h2 variants: bash tools/h2_regs.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I style_transfer_amd/csrc -DSTX_H2_DEV_SUBSET "$@" \
  --offload-device-only -S -Rpass-analysis=kernel-resource-usage style_transfer_amd/csrc/conv_h2.hip -o /tmp/h2_dev.s > /tmp/h2_dev.log 2>&1
grep -E "error" -A4 /tmp/h2_dev.log | head -20
python tools/regs.py /tmp/h2_dev.log conv_h2
