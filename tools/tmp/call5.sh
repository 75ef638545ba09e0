cd /root/repo
timeout 900 python -m pytest tests/test_gpu_end_to_end.py -q -x -k "lbfgs" -s 2>&1 | grep "final image\|nearest\|per-pixel\|passed\|failed" | tail -6
