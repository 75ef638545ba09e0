cd /root/repo
timeout 900 python tools/tmp/lbfgs_spread.py 2>&1 | tail -20
