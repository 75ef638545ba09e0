cd /root/repo
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/full/pytest_tail.txt
cat gpurun_out/full/pytest_tail.txt
python __graft_entry__.py --smoke 2>&1 | tail -3
