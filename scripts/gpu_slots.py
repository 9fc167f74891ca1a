"""k_solve on the bench shape with a given number of resident workgroups (n_slots; 0 = what the occupancy query allows): does a workgroup run faster when
fewer share the memory system?   python scripts/gpu_slots.py <n_scen> <n_slots> [<n_slots> ...]"""
import sys, time
sys.path.insert(0, '.')
import bench
from pyhybridcontrol_amd import gpu, host
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 64
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents])
for ns in [int(a) for a in sys.argv[2:]] or [0]:
    prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=800, max_pivots=40000, n_slots=ns)
    prob.upload(x0, om, midx); st = prob.solve_resident(); st = prob.solve_resident()
    print("n_slots %4d (asked %d)  solve_ms %8.1f  pivots %d  slot-ms per instance %.3f" % (prob.opts.n_slots, ns, st["solve_ms"], st["pivots"], st["solve_ms"] * prob.opts.n_slots / x0.shape[0]), flush=True)
    prob.close()
