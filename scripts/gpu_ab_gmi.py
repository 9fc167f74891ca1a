"""A/B of the wave-parallel Gomory round against the one-at-a-time version (solver option reserved bit 5) on the bench shard."""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from pyhybridcontrol_amd import gpu, host
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 512, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents])
res = {}
for flag in (0, 32):
    prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=400, max_pivots=20000, reserved=flag)
    prob.upload(x0, om, midx); prob.solve_resident(); st = prob.solve_resident()
    out = prob.download(); res[flag] = out
    print('flag', flag, '%.3f s %.0f/s' % (st['solve_ms'] / 1e3, x0.shape[0] / st['solve_ms'] * 1e3), 'proven %.2f%%' % (100 * (out['status'] == 0).mean()), 'cuts', st['cuts'], 'pivots', st['pivots'], flush=True)
    prob.close()
a, b = res[0], res[32]
print('status equal', (a['status'] == b['status']).mean(), 'obj rel diff max', np.nanmax(np.abs(a['obj'] - b['obj']) / np.maximum(1, np.abs(b['obj']))))
both = (a['status'] == 0) & (b['status'] == 0)
rd = np.abs(a['obj'] - b['obj']) / np.maximum(1, np.abs(b['obj']))
print('both proven', int(both.sum()), 'max rel diff among them %.4f' % rd[both].max(), '(must be within the 1 % gap)')
worst = np.argsort(-np.where(np.isfinite(rd), rd, 0))[:5]
print('largest differences:', [(int(i), int(a['status'][i]), int(b['status'][i]), float(a['obj'][i]), float(b['obj'][i]), float(a['lower_bound'][i]), float(b['lower_bound'][i])) for i in worst])
