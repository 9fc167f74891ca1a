import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 1, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
ms = [model.condense_device(N_t) for _ in range(10)]
print('condense ms', min(ms), sorted(ms)[len(ms)//2])
