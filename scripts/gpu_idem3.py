import sys, numpy as np
sys.path.insert(0,'.')
from pyhybridcontrol_amd import gpu, host, synthetic as syn
i=int(sys.argv[1])
wl=syn.make_workload('cfg3', batch=1024); ag=wl['agents'][0]; d=ag['dims']
m=gpu.GpuModel([ag['mats']], d)
p=gpu.GpuProblem(m, wl['N_p'], wl['N_tilde'], host.cost_from_atoms(ag['atoms'], d, wl['N_p'], wl['N_tilde']), max_nodes=300, reserved=int(sys.argv[2]) if len(sys.argv)>2 else 0)
for t in range(4):
    r=p.solve(ag['x0'][i:i+1], ag['omega'][i:i+1])
    print(t, r['obj'][0], r['status'][0], r['nodes'][0], r['pivots'][0], 'refactors', r['stats']['refactors'], 'cuts', r['stats']['cuts'])
