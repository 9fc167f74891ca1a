import sys, numpy as np
sys.path.insert(0,'.')
import bench
from pyhybridcontrol_amd import gpu, host
ids=[int(a) for a in sys.argv[1:]]
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, max(ids)//64+1, 0)
d=agents[0]['dims']
model=gpu.GpuModel([a['mats'] for a in agents], d)
prob=gpu.GpuProblem(model, N_p, N_t, host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents]), gap_rel=1e-2, max_nodes=400, max_pivots=20000, reserved=int(__import__('os').environ.get('MLD_DEBUG','0')))
for i in ids:
    out=prob.solve(x0[i:i+1], om[i:i+1], midx[i:i+1])
    print(i, 'alone: status', out['status'][0], 'obj', out['obj'][0], 'lb', out['lower_bound'][0], 'nodes', out['nodes'][0], 'pivots', out['pivots'][0], 'refac', out['stats']['refactors'], 'cuts', out['stats']['cuts'])
sel=np.array(ids)
out=prob.solve(x0[sel], om[sel], midx[sel])
print('together', out['status'], out['obj'], out['nodes'], out['pivots'])
