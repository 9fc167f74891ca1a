import sys, os, numpy as np
sys.path.insert(0,'.')
from pyhybridcontrol_amd import gpu, host, synthetic as syn
name=sys.argv[1]; nb=int(sys.argv[2])
wl=syn.make_workload(name, batch=nb); ag=wl['agents'][0]; d=ag['dims']
m=gpu.GpuModel([ag['mats']], d)
p=gpu.GpuProblem(m, wl['N_p'], wl['N_tilde'], host.cost_from_atoms(ag['atoms'], d, wl['N_p'], wl['N_tilde']), max_nodes=2000, reserved=int(os.environ.get('MLD_DEBUG','0')))
out=p.solve(ag['x0'], ag['omega'])
print(out['status'], out['obj'], out['nodes'], out['pivots'])
