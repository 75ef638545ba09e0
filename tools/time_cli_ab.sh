#!/bin/bash
# Same-box A/B of the whole `--size 2048 --tile-size 1024` run under an environment switch:
#   bash tools/time_cli_ab.sh STX_GRAPH 1 0 [repeats]        (a value `unset` leaves the variable unset)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
VAR=$1; A=$2; B=$3; N=${4:-2}
python "$REPO/tools/make_inputs.py" /tmp/stx_in 2048 >/dev/null
cd /tmp
for i in $(seq $N); do
  for v in $A $B; do
    if [ "$v" = unset ]; then SET="-u $VAR"; else SET="$VAR=$v"; fi
    env $SET python "$REPO/style_transfer.py" -ci /tmp/stx_in/content.png -si /tmp/stx_in/style.png --devices 0 \
        --weights synthetic --size 2048 --tile-size 1024 -oi /tmp/stx_out_ab.png 2>&1 | tail -2 | tr '\n' ' '
    echo " [$VAR=$v]"
  done
done
