import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu
n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 64
agents, N_p, N_t, x0, om, midx = bench.make_shard(n_models, 1, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
ms = [model.condense_device(N_t) for _ in range(10)]
print("models", n_models, "condense ms min", min(ms), "median", sorted(ms)[len(ms)//2], "TB/s at min %.2f" % (n_models * 4393696 / min(ms) / 1e9))
