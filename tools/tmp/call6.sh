cd /root/repo
python tools/profile_layers.py 965 6 2>&1 | tail -60
