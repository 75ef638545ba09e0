#!/bin/bash
# Wall-clock of whole CLI runs on synthetic pictures (used for the numbers in DESIGN.md section 5).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
python "$REPO/tools/make_inputs.py" /tmp/stx_in 2048 >/dev/null
cd /tmp
run() {
    local label=$1; shift
    local s=$(date +%s%N)
    python "$REPO/style_transfer.py" -ci /tmp/stx_in/content.png -si /tmp/stx_in/style.png --devices 0 --weights synthetic "$@" 2>&1 | tail -2
    local e=$(date +%s%N)
    echo "$label: wall $(( (e - s) / 1000000 )) ms"
}
run "adam --size 1024 --tile-size 1024" --size 1024 --tile-size 1024 -oi /tmp/stx_out_1024.png
run "adam --size 2048 --tile-size 1024" --size 2048 --tile-size 1024 -oi /tmp/stx_out_2048.png
run "lbfgs --size 2048 --tile-size 1024 -i 100" --size 2048 --tile-size 1024 -o lbfgs -i 100 -oi /tmp/stx_out_l.png
if [ "$1" = "all" ]; then
    python "$REPO/tools/make_inputs.py" /tmp/stx_in 4096 >/dev/null
    run "config 4: lbfgs --size 4096 --tile-size 1024" --size 4096 --tile-size 1024 -o lbfgs -oi /tmp/stx_out_c4.png
    python "$REPO/tools/make_inputs.py" /tmp/stx_in 2048 >/dev/null      # (2048^2 pictures again: decoding three 4096^2 PNGs is 0.9 s of its own)
    run "config 5: vgg16_avgpool, two styles, --size 2048" --size 2048 --tile-size 1024 --model vgg16_avgpool.prototxt -si /tmp/stx_in/style.png /tmp/stx_in/content.png -oi /tmp/stx_out_c5.png
fi
