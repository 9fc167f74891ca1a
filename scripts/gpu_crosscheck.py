"""Cross-check of the bench workload: GPU vs the C oracle on the first N instances of the cfg4 shard (MIPGap 1e-2,
NodeLimit 400): status agreement, gap-consistent objectives, certificates of the GPU points in the original rows."""
import os, sys, time, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import bench, condense_np as cn, orc, tighten_np
from pyhybridcontrol_amd import gpu, host
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, (N + 63) // 64, 0)
x0, om, midx = x0[:N], om[:N], midx[:N]
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
prob = gpu.GpuProblem(model, N_p, N_t, host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents]),
                      gap_rel=1e-2, max_nodes=400, max_pivots=20000)
out = prob.solve(x0, om, midx)
forms, raw = {}, {}
agree = both_opt = cons = 0
worst_rel = 0.0; viol = 0.0; bad = []
t0 = time.time()
for i in range(N):
    a = int(midx[i]); ag = agents[a]
    if a not in forms:
        forms[a] = cn.standard_form(tighten_np.tighten(ag['mats'], d, nu_l=d['nu_l']), ag['atoms'], N_p, N_t, nu_l=d['nu_l'])
        raw[a] = cn.standard_form(ag['mats'], ag['atoms'], N_p, N_t, nu_l=d['nu_l'])
    sf, sf0 = forms[a], raw[a]
    h = cn.rhs(sf['evo'], x0[i], om[i]); q = cn.lin_cost(sf['cost'], x0[i], om[i])
    r0 = cn.cost_const(sf['cost']['const_terms'], x0[i], om[i])
    ref = orc.solve_milp(q, sf['G'], h, sf['lb'], sf['ub'], sf['is_bin'], gap_rel=1e-2, max_nodes=400, presolve=4, max_pivots=20000)
    gs = gpu._lib.STATUS_NAMES[int(out['status'][i])]
    agree += gs == ref['status']
    if np.isfinite(out['obj'][i]):
        v = out['v'][i]; G0 = sf0['G']
        rown = np.maximum(1.0, np.abs(G0).max(axis=1))
        viol = max(viol, float(((G0 @ v - cn.rhs(sf0['evo'], x0[i], om[i])) / rown).max()))
        assert np.all((v[sf0['is_bin']] == 0) | (v[sf0['is_bin']] == 1))
        assert abs(q @ v + r0 - out['obj'][i]) <= 1e-6 * max(1.0, abs(out['obj'][i]))
    if gs == 'optimal' and ref['status'] == 'optimal':
        both_opt += 1
        o1, o2 = out['obj'][i], ref['obj'] + r0
        rel = abs(o1 - o2) / max(1e-9, abs(min(o1, o2)))
        worst_rel = max(worst_rel, rel)
        # both are within 1 % of the optimum, so they are within ~1 % of each other
        if rel <= 1.02e-2: cons += 1
        else: bad.append((i, o1, o2))
print('instances', N, 'status agreement', agree, 'both optimal', both_opt, 'gap-consistent', cons, 'worst relative difference %.3e' % worst_rel,
      'max scaled row violation of GPU points %.2e' % viol, 'oracle s', round(time.time() - t0, 1))
print('GPU status', {k: int((out['status'] == s).sum()) for s, k in enumerate(gpu._lib.STATUS_NAMES)}, 'inconsistent', bad[:5])
