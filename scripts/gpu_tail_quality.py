"""The node-limited tail of the bench shard: distribution of the reported gap, and how far the incumbents are from HiGHS optima."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import bench
from pyhybridcontrol_amd import gpu, host
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 512
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents])
prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=400, max_pivots=20000)
out = prob.solve(x0, om, midx)
nl = np.where(out['status'] == 2)[0]
gap = (out['obj'][nl] - out['lower_bound'][nl]) / np.maximum(1e-9, np.abs(out['obj'][nl]))
print('node-limited', nl.size, 'gap percentiles 50/75/90/99/max', np.round(np.percentile(gap, [50, 75, 90, 99, 100]), 4))
print('reported gap > 10 %:', int((gap > 0.1).sum()), ' > 50 %:', int((gap > 0.5).sum()))
try:
    import condense_np as cn
    from scipy.optimize import milp, LinearConstraint, Bounds
    worst = nl[np.argsort(-gap)[:12]]
    forms = {}
    for i in worst:
        a = int(midx[i]); ag = agents[a]
        if a not in forms: forms[a] = cn.standard_form(ag['mats'], ag['atoms'], N_p, N_t, nu_l=ag['dims']['nu_l'])
        sf = forms[a]
        h, q = cn.rhs(sf['evo'], x0[i], om[i]), cn.lin_cost(sf['cost'], x0[i], om[i])
        r = cn.cost_const(sf['cost']['const_terms'], x0[i], om[i])
        ref = milp(q, constraints=LinearConstraint(sf['G'], -np.inf, h), bounds=Bounds(sf['lb'], sf['ub']), integrality=sf['is_bin'].astype(int), options=dict(mip_rel_gap=1e-3, time_limit=20))
        print('inst %5d gpu obj %.4f lb %.4f | highs %.4f' % (i, out['obj'][i], out['lower_bound'][i], ref.fun + r if ref.x is not None else np.nan), flush=True)
except Exception as e:
    print('highs comparison skipped:', e)
