"""throughput of the bench workload (16384 instances) against the number of solver slots (persistent workgroups)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu, host
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 256, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents])
for k in (128, 192, 224, 256, 320):
    p = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=400, max_pivots=20000, n_slots=k)
    p.upload(x0, om, midx); p.solve_resident(); st = p.solve_resident()
    print(k, 'solve_ms %.1f' % st['solve_ms'], '%.0f/s' % (x0.shape[0] / st['solve_ms'] * 1e3), flush=True)
    p.close()
