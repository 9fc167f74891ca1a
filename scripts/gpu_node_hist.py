"""Where the node budget goes on the bench shard: nodes used by proven instances, and what a smaller NodeLimit would lose."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu, host
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 512
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents])
for lim in (400, 300, 200, 150, 100):
    prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=lim, max_pivots=20000)
    prob.upload(x0, om, midx); prob.solve_resident(); st = prob.solve_resident()
    out = prob.download(); tel = prob.telemetry()
    ok = out['status'] == 0; nl = out['status'] == 2
    gap = (out['obj'][nl] - out['lower_bound'][nl]) / np.maximum(1e-9, np.abs(out['obj'][nl])) if 'lower_bound' in out else np.array([np.nan])
    print('NodeLimit %3d: %.3f s  %.0f/s  proven %.2f%%  node-limited %d (median gap %.2f%%)  no-incumbent %d  nodes of proven p50/p90/p99/max %s' % (
        lim, st['solve_ms'] / 1e3, x0.shape[0] / st['solve_ms'] * 1e3, 100 * ok.mean(), int(nl.sum()), 100 * np.median(gap) if gap.size else 0,
        int((~np.isfinite(out['obj'])).sum()), np.percentile(out['nodes'][ok], [50, 90, 99, 100]).astype(int)), flush=True)
    if lim == 400:
        nd = out['nodes'][ok]
        print('   proven with more than 100/150/200/300 nodes:', [(int((nd > t).sum())) for t in (100, 150, 200, 300)])
    prob.close()
