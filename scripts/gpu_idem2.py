import sys, numpy as np
sys.path.insert(0,'.')
from pyhybridcontrol_amd import gpu, host, synthetic as syn
nb=int(sys.argv[1])
wl=syn.make_workload('cfg3', batch=nb); ag=wl['agents'][0]; d=ag['dims']
m=gpu.GpuModel([ag['mats']], d)
cost=host.cost_from_atoms(ag['atoms'], d, wl['N_p'], wl['N_tilde'])
def run(**kw):
    p=gpu.GpuProblem(m, wl['N_p'], wl['N_tilde'], cost, max_nodes=300, **kw)
    r=p.solve(ag['x0'], ag['omega']); p.close(); return r
a=run(n_slots=1); b=run(n_slots=1); c=run(n_slots=8); e=run(n_slots=8)
def diff(x,y): return int((x['pivots']!=y['pivots']).sum()), int((x['obj']!=y['obj']).sum())
print('1 vs 1', diff(a,b), '8 vs 8', diff(c,e), '1 vs 8', diff(a,c))
# each instance alone in a fresh problem
p=gpu.GpuProblem(m, wl['N_p'], wl['N_tilde'], cost, max_nodes=300, n_slots=1)
alone=[p.solve(ag['x0'][i:i+1], ag['omega'][i:i+1]) for i in range(nb)]
print('alone vs slots=1 pivots differ:', [i for i in range(nb) if alone[i]['pivots'][0]!=a['pivots'][i]])
again=[p.solve(ag['x0'][i:i+1], ag['omega'][i:i+1]) for i in range(nb)]
print('alone twice differ:', [i for i in range(nb) if alone[i]['pivots'][0]!=again[i]['pivots'][0]])
