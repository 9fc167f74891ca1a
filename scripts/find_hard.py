import sys, numpy as np
sys.path.insert(0,'.')
import bench
from pyhybridcontrol_amd import gpu, host
n_scen=int(sys.argv[1]); maxp=int(sys.argv[2])
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d=agents[0]['dims']
model=gpu.GpuModel([a['mats'] for a in agents], d)
prob=gpu.GpuProblem(model, N_p, N_t, host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents]), gap_rel=1e-2, max_nodes=400, max_pivots=maxp)
out=prob.solve(x0, om, midx)
print('solve_ms', out['stats']['solve_ms'], {k:out['stats'][k] for k in ('n_optimal','n_infeasible','n_node_limit','n_numerical')})
piv=out['pivots']; order=np.argsort(-piv)[:12]
print('top pivots', [(int(i), int(piv[i]), int(out['nodes'][i]), int(out['status'][i])) for i in order])
print('status1', np.where(out['status']==1)[0][:10], 'status3', np.where(out['status']==3)[0][:10], 'noinc', np.where(~np.isfinite(out['obj']))[0][:10])
print('pivot percentiles', np.percentile(piv,[50,90,99,99.9,100]))
