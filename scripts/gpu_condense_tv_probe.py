"""k_condense_tv (one model per horizon step) beside K1+K2 (k_condense_model + k_condense_flat) on the bench shard's models:
kernel milliseconds (HIP events inside the library) and algorithmic GB/s (outputs written + step models read)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import bench
import _tv
from pyhybridcontrol_amd import gpu

n_agents = int(sys.argv[1]) if len(sys.argv) > 1 else 64
agents, N_p, N_t, x0, om, midx = bench.make_shard(n_agents, 1, 0)
d = agents[0]['dims']
shapes = gpu.evo_shapes(d, N_t)
out_bytes = 8 * sum(r * c for r, c in shapes.values())
in_bytes = 8 * sum(np.prod(gpu.mat_shape(nm, d)) for nm in gpu.MAT_NAMES)
lti = gpu.GpuModel([a['mats'] for a in agents], d)
ms = sorted(lti.condense_device(N_t) for _ in range(10))
alg = n_agents * (out_bytes + in_bytes)
print('LTI  n=%d N=%d  ms min/med %.4f %.4f  alg %.1f MB  %.0f GB/s' % (n_agents, N_t, ms[0], ms[5], alg / 1e6, alg / ms[5] / 1e6))
lti.close()
tv = gpu.GpuModel([_tv.step_models(a['mats'], N_t, seed=i) for i, a in enumerate(agents)], d, time_varying=True)
ms = sorted(tv.condense_device(N_t) for _ in range(10))
alg = n_agents * (out_bytes + N_t * in_bytes)
print('TV   n=%d N=%d  ms min/med %.4f %.4f  alg %.1f MB  %.0f GB/s' % (n_agents, N_t, ms[0], ms[5], alg / 1e6, alg / ms[5] / 1e6))
tv.close()
