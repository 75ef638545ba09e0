"""Functional check of the multi-GPU legs on whatever HIP devices are visible (manual tool, never
collected by pytest, never run by bench.py on its own):

    python tools/multi_device_check.py [n_devices]

1. TileFarm over devices 0..n-1 in ONE host process (hipDeviceEnablePeerAccess, cross-device
   hipMemcpyAsync in both directions, stx_engine_wait across devices): bench.FarmJob on the
   metric's 2048 x 2048 / four-tile workload -- step 1 on all devices against device 0 alone, bit
   for bit, then a few timed steps;
2. the same for config 4's 4096 x 4096 / 16-tile grid (several tiles per device, staging slots);
prints one JSON record per leg.  The one-process-per-GPU leg over RCCL is `python -m
torch.distributed.run ... bench.py --gpus N` itself (tools/partition_probe.sh runs both).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from style_transfer_amd import lib

n_vis = lib.device_count()
n = int(sys.argv[1]) if len(sys.argv) > 1 else n_vis
print(json.dumps({'visible_devices': n_vis, 'names': [lib.device_name(d) for d in range(n_vis)],
                  'using': n}), flush=True)
if n_vis < n or n < 2:
    sys.exit('need at least 2 visible HIP devices (have %d, asked for %d)' % (n_vis, n))
devices = list(range(n))
for rows, cols, opt, steps in ((2, 2, 'adam', 5), (4, 4, 'lbfgs', 3)):
    rec = bench.farm_leg(devices, rows, cols, steps, 1, opt)
    print(json.dumps(rec), flush=True)
    if rec['bit_identical'] is not True:
        sys.exit('NOT bit-identical on %d devices' % n)
print('multi_device_check: OK')
