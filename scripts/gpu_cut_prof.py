"""GPU (build with MLD_CXXFLAGS=-DMLD_CUT_PROF): split of the cut-separation clock of k_solve on the bench shard -- c-MIR scoring (phase A),
c-MIR build (phase B), Gomory rounds -- as shares of the workgroup time."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import bench
from pyhybridcontrol_amd import gpu, host, _lib
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 512
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
prob = gpu.GpuProblem(model, N_p, N_t, host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents]), gap_rel=1e-2, max_nodes=800, max_pivots=40000)
prob.upload(x0, om, midx); prob.solve_resident(); st = prob.solve_resident()
out = (C.c_int64 * 8)(); _lib.load().mld_debug_profile(prob._h, out)
t = np.array(list(out), dtype=float)
tot = t[0] + t[1] + t[2] + t[5] + t[6]       # (slots 3 / 4 / 7 carry the cut split in this build; leaf / bound-change time is then not in the total)
print("solve_ms", st["solve_ms"], "cuts share of (update+select+cuts+verify+setup): %.3f" % (t[2] / tot))
print("c-MIR scoring %.3f  c-MIR build %.3f  Gomory %.3f  (of the cut clock)" % (t[3] / t[2], t[4] / t[2], t[7] / t[2]))
