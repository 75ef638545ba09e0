cd /root/repo
mkdir -p gpurun_out/c4
for st in "" "8 10" "8 20" "8 40" "4 10" "4 20" "16 10" "16 20"; do
  if [ -z "$st" ]; then unset STX_H2_STAGGER; else export STX_H2_STAGGER="$st"; fi
  echo "== stagger '$st'"
  python tools/profile_layers.py 1024 6 2>&1 | grep -E "conv1_2|conv2_1|conv2_2|conv3_2 |TOTAL" | head -9
done
unset STX_H2_STAGGER
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-wall-clock --steady-seconds 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('off', round(d['value'],1), round(d['steady']['value'],1))"
  STX_H2_STAGGER="8 20" python bench.py --no-cpu-baseline --no-wall-clock --steady-seconds 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8 20', round(d['value'],1), round(d['steady']['value'],1))"
done
