"""ad-hoc GPU probe: solve a synthetic batch, print throughput and solver statistics"""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from pyhybridcontrol_amd import gpu, host, synthetic as syn

name = sys.argv[1]; batch = int(sys.argv[2]); n_agents = int(sys.argv[3]) if len(sys.argv) > 3 else 1
opts = dict(a.split('=') for a in sys.argv[4:])
opts = {k: float(v) if '.' in v or 'e' in v else int(v) for k, v in opts.items()}
wl = syn.make_workload(name, batch=batch, n_agents=n_agents)
d = wl['agents'][0]['dims']
t0 = time.time()
m = gpu.GpuModel([a['mats'] for a in wl['agents']], d)
cost = host.stack_costs([host.cost_from_atoms(a['atoms'], d, wl['N_p'], wl['N_tilde']) for a in wl['agents']])
p = gpu.GpuProblem(m, wl['N_p'], wl['N_tilde'], cost, **opts)
t1 = time.time()
x0 = np.concatenate([a['x0'] for a in wl['agents']]); om = np.concatenate([a['omega'] for a in wl['agents']])
midx = np.repeat(np.arange(n_agents), batch).astype(np.int32)
p.upload(x0, om, midx)
st = p.solve_resident()
out = p.download()
nb = x0.shape[0]
print(json.dumps(dict(name=name, instances=nb, create_s=round(t1 - t0, 3), solve_ms=round(st['solve_ms'], 2), rhs_ms=round(st['rhs_ms'], 3),
                      inst_per_s=round(nb / st['solve_ms'] * 1e3, 1), nodes=st['nodes'], pivots=st['pivots'], cuts=st['cuts'], refactors=st['refactors'],
                      n_opt=st['n_optimal'], n_inf=st['n_infeasible'], n_lim=st['n_node_limit'], n_num=st['n_numerical'],
                      pivots_per_s=round(st['pivots'] / st['solve_ms'] * 1e3), max_pivots=int(out['pivots'].max()), med_pivots=int(np.median(out['pivots'])))))
