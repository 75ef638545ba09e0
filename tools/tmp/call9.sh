cd /root/repo
python tools/profile_cli.py 2048 2>&1 | head -75
