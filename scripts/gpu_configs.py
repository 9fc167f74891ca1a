"""BASELINE configs 1-3 beside the bench line: cfg2 relaxation only (binaries fixed), cfg2 and cfg3 full branch-and-cut
(MILP, and MIQP with the Q_x deviation cost), exact gap.  Prints one JSON line per case."""
import os, sys, json, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyhybridcontrol_amd import gpu, host, synthetic as syn


def run(name, batch, quadratic=False, relax=False, **opts):
    wl = syn.make_workload(name, batch=batch, quadratic=quadratic)
    ag = wl['agents'][0]; d = ag['dims']
    m = gpu.GpuModel([ag['mats']], d)
    p = gpu.GpuProblem(m, wl['N_p'], wl['N_tilde'], host.cost_from_atoms(ag['atoms'], d, wl['N_p'], wl['N_tilde']), **opts)
    fixed = None
    if relax:   # binaries fixed at the values of a full solve: the LP / QP relaxation kernel only
        full = p.solve(ag['x0'], ag['omega'])
        is_bin = p.is_bin
        fixed = np.where(np.isfinite(full['obj'])[:, None], np.rint(full['v'][:, is_bin]), 0).astype(np.uint8)
    p.upload(ag['x0'], ag['omega'], None, fixed); p.solve_resident(); st = p.solve_resident(); out = p.download()
    print(json.dumps(dict(case="%s%s%s batch %d" % (name, " MIQP" if quadratic else " MILP", " relaxation-only" if relax else "", batch),
                          n=p.n, binaries=p.n_bin, solve_ms=round(st['solve_ms'], 2), solves_per_s=round(batch / st['solve_ms'] * 1e3),
                          optimal=st['n_optimal'], infeasible=st['n_infeasible'], node_limit=st['n_node_limit'], numerical=st['n_numerical'],
                          nodes_per_inst=round(st['nodes'] / batch, 1), pivots_per_inst=round(st['pivots'] / batch, 1))), flush=True)
    p.close(); m.close()


run("cfg2", 256, relax=True)
run("cfg2", 256, quadratic=True, relax=True)
run("cfg2", 256, gap_rel=1e-2, max_nodes=400)
run("cfg3", 1024, gap_rel=1e-2, max_nodes=400)
run("cfg3", 1024, quadratic=True, gap_rel=1e-2, max_nodes=400)
