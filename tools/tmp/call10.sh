cd /root/repo
B=tools/ubench/bin/h2conv_bench
for hw in "181 181" "182 182" "184 184" "91 91" "92 92" "96 96"; do
  for d in 0 1; do
    if [ "$hw" = "91 91" ] || [ "$hw" = "92 92" ] || [ "$hw" = "96 96" ]; then K=512; else K=256; fi
    $B $K $K $hw 2 $d 2>&1 | grep -E "^(fwd|bwd)"
  done
done
