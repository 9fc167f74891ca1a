import sys, numpy as np
sys.path.insert(0,'.')
from pyhybridcontrol_amd import gpu, host, synthetic as syn
nb=int(sys.argv[1])
wl=syn.make_workload('cfg3', batch=nb); ag=wl['agents'][0]; d=ag['dims']
m=gpu.GpuModel([ag['mats']], d)
p=gpu.GpuProblem(m, wl['N_p'], wl['N_tilde'], host.cost_from_atoms(ag['atoms'], d, wl['N_p'], wl['N_tilde']), max_nodes=300)
a=p.solve(ag['x0'], ag['omega']); b=p.solve(ag['x0'], ag['omega'])
dv=np.where(np.any(a['v']!=b['v'],axis=1))[0]; do=np.where(a['obj']!=b['obj'])[0]; ds=np.where(a['status']!=b['status'])[0]
print('solve_ms', a['stats']['solve_ms'], b['stats']['solve_ms'])
print('diff v', len(dv), dv[:10], 'diff obj', len(do), do[:10], 'diff status', len(ds))
for i in do[:5]: print(i, a['obj'][i], b['obj'][i], a['status'][i], b['status'][i], a['nodes'][i], b['nodes'][i], a['pivots'][i], b['pivots'][i])
dp=np.where(a['pivots']!=b['pivots'])[0]; print('diff pivots', len(dp), dp[:10])
