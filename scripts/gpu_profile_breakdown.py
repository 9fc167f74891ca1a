import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import bench
from pyhybridcontrol_amd import gpu, host, _lib
n_scen=int(sys.argv[1]); nodes=int(sys.argv[2]) if len(sys.argv)>2 else 400
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d=agents[0]['dims']
model=gpu.GpuModel([a['mats'] for a in agents], d)
prob=gpu.GpuProblem(model, N_p, N_t, host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents]), gap_rel=1e-2, max_nodes=nodes, max_pivots=20000)
prob.upload(x0, om, midx); st=prob.solve_resident(); st=prob.solve_resident()
out=(C.c_int64*8)(); _lib.load().mld_debug_profile(prob._h, out)
tel=prob.telemetry(); tot=tel['latency_ns'].sum()
names=['pivot_update','simplex_select','cuts','leaf','set_bounds','residual/refactor','setup','-']
ticks=np.array(list(out),dtype=float); 
print('solve_ms',st['solve_ms'],'pivots',st['pivots'],'nodes',st['nodes'],'inst',x0.shape[0],'refactors',st['refactors'],'cuts',st['cuts'])
for n_,t in zip(names,ticks): print('%-18s %6.1f%%'%(n_, 100*t/ticks[:7].sum() if n_!='-' else 0))
print('avg nnz(pivot row)', ticks[7]/st['pivots'], 'of', prob.n+1, '; avg rows touched', tel['rows_updated'].sum()/st['pivots'])
print('sum-of-latency s', tot*1e-9, 'ticks total (100MHz?) s', ticks[:7].sum()/1e8, ' per pivot us (pivot_update)', ticks[0]/1e8/st['pivots']*1e6)
out=prob.download(); lat=tel['latency_ns']*1e-9
for s_,nm in ((0,'optimal'),(2,'node_limit')):
    sel=out['status']==s_
    print(nm, 'count', int(sel.sum()), 'share of WG-time %.1f%%'%(100*lat[sel].sum()/lat.sum()), 'mean ms %.1f'%(1e3*lat[sel].mean() if sel.any() else 0), 'mean pivots %.0f'%(out['pivots'][sel].mean() if sel.any() else 0))
