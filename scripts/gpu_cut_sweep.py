"""Throughput / proven share of the bench shard against the root cut loop's parameters."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu, host
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 512
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents])
configs = [dict(), dict(cut_rounds=4), dict(cut_rounds=6), dict(cut_rounds=12), dict(mir_per_round=10), dict(mir_per_round=40),
           dict(cuts_per_round=20), dict(cuts_per_round=80, max_cuts=300), dict(cut_rounds=6, mir_per_round=30)]
if len(sys.argv) > 2 and sys.argv[2] == "combo":
    configs = [dict(), dict(cuts_per_round=80, max_cuts=300, cut_rounds=10, mir_per_round=10), dict(cuts_per_round=80, max_cuts=300, cut_rounds=10, mir_per_round=15),
               dict(cuts_per_round=80, max_cuts=300, cut_rounds=12, mir_per_round=10), dict(cuts_per_round=80, max_cuts=400, cut_rounds=12, mir_per_round=12)]
elif len(sys.argv) > 2 and sys.argv[2] == "wide":     # more Gomory cuts (cheap since they are derived eight at a time), fewer rounding cuts
    configs = [dict(), dict(cuts_per_round=80, max_cuts=300), dict(cuts_per_round=80, max_cuts=300, mir_per_round=10),
               dict(cuts_per_round=80, max_cuts=300, mir_per_round=15), dict(cuts_per_round=120, max_cuts=400),
               dict(cuts_per_round=120, max_cuts=400, mir_per_round=10), dict(cuts_per_round=80, max_cuts=300, cut_rounds=10)]
if len(sys.argv) > 2 and sys.argv[2] == "r2":      # round 2 (penalty branching, NodeLimit 800): the deep settings round 1 withdrew
    configs = [dict(), dict(mir_per_round=10), dict(cuts_per_round=80, max_cuts=300, cut_rounds=10, mir_per_round=10),
               dict(cuts_per_round=80, max_cuts=400, cut_rounds=12, mir_per_round=12), dict(cuts_per_round=80, max_cuts=400, cut_rounds=12, mir_per_round=20),
               dict(cut_rounds=6), dict(cut_rounds=6, mir_per_round=10)]
nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 400
for kw in configs:
    prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=nodes, max_pivots=40000, **kw)
    prob.upload(x0, om, midx); prob.solve_resident(); st = prob.solve_resident()
    out = prob.download()
    ok = out['status'] == 0
    print('%-45s %.3f s  %.0f/s  proven %.2f%%  pivots/inst %.0f nodes/inst %.1f' % (kw, st['solve_ms'] / 1e3, x0.shape[0] / st['solve_ms'] * 1e3,
          100 * ok.mean(), out['pivots'].mean(), out['nodes'].mean()), flush=True)
    prob.close()
