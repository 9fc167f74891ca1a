"""the shards ranks 1..7 of an 8-GPU run would get (same generator, other scenario offsets), solved one after another on
this GPU: status mix per shard (guards against surprises in the multi-GPU bench)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu, host
model = prob = None
for rank in range(1, 8):
    agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 512, rank * 512)
    d = agents[0]['dims']
    if model is None:
        model = gpu.GpuModel([a['mats'] for a in agents], d)
        prob = gpu.GpuProblem(model, N_p, N_t, host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents]), gap_rel=1e-2, max_nodes=400, max_pivots=20000)
    prob.upload(x0, om, midx); st = prob.solve_resident(); out = prob.download()
    print('rank', rank, 'solve_ms %.0f' % st['solve_ms'], {k: st[k] for k in ('n_optimal', 'n_infeasible', 'n_node_limit', 'n_numerical')},
          'no incumbent', int((~np.isfinite(out['obj'])).sum()), flush=True)
