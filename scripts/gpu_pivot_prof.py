"""sub-phase timing of s_pivot (library built with MLD_CXXFLAGS=-DMLD_PIVOT_PROF)"""
import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import bench
from pyhybridcontrol_amd import gpu, host, _lib
n_scen=int(sys.argv[1])
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d=agents[0]['dims']
model=gpu.GpuModel([a['mats'] for a in agents], d)
prob=gpu.GpuProblem(model, N_p, N_t, host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents]), gap_rel=1e-2, max_nodes=400, max_pivots=20000)
prob.upload(x0, om, midx); st=prob.solve_resident(); st=prob.solve_resident()
out=(C.c_int64*8)(); _lib.load().mld_debug_profile(prob._h, out)
t=np.array(list(out),dtype=float)
names=['stage row/col','sector list','xB,row writeback,rowlist','update loop','cost row+bookkeeping']
print('solve_ms',st['solve_ms'],'pivots',st['pivots'])
for n_,v in zip(names,t[:5]): print('%-28s %6.2f us/pivot'%(n_, v/1e8/st['pivots']*1e6))
