cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "style_terms_over" 2>&1 | tail -15
