"""paired comparison, instance by instance, of the kernel and the oracle on the first n instances of the bench shard (bench options): nodes, pivots, status.
The oracle's side comes from scripts/cpu_study.py (an .npz made on the CPU box and passed in: it travels with the repo snapshot).
    python scripts/gpu_vs_oracle_paired.py <oracle.npz> [n=2048]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import bench
from pyhybridcontrol_amd import gpu, host
z = np.load(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, (n + 63) // 64, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
prob = gpu.GpuProblem(model, N_p, N_t, host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents]), gap_rel=1e-2, max_nodes=800, max_pivots=40000)
out = prob.solve(x0[:n], om[:n], midx[:n])
gn, gp, gs = out["nodes"].astype(float), out["pivots"].astype(float), out["status"]
on, op, os_ = z["nd"][:n], z["pv"][:n], z["st"][:n]
print("GPU   : nodes/inst %.2f pivots/inst %.1f node-limited %d cuts/inst %.1f" % (gn.mean(), gp.mean(), (gs == 2).sum(), out["stats"]["cuts"] / n))
print("oracle: nodes/inst %.2f pivots/inst %.1f node-limited %d cuts/inst %.1f" % (on.mean(), op.mean(), (os_ == 2).sum(), z["ct"][:n].mean()))
r = gp / np.maximum(1, op)
print("pivot ratio GPU/oracle: geometric mean %.3f median %.3f; GPU > 3x oracle: %d, oracle > 3x GPU: %d" % (np.exp(np.log(np.maximum(r, 1e-3)).mean()), np.median(r), (r > 3).sum(), (r < 1 / 3).sum()))
one = (on <= 1)
print("instances the oracle proves at its root (%d): GPU nodes mean %.2f, GPU also at the root %d" % (one.sum(), gn[one].mean(), (gn[one] <= 1).sum()))
gone = (gn <= 1)
print("instances the GPU proves at its root (%d): oracle nodes mean %.2f, oracle also at the root %d" % (gone.sum(), on[gone].mean(), (on[gone] <= 1).sum()))
opt = z["opt"][:n]
rbg = out["lower_bound"]
worst = np.argsort(-(gp - op))[:10]
for i in worst: print("  inst %4d GPU nodes %4d piv %5d st %d | oracle nodes %4d piv %5d st %d" % (i, gn[i], gp[i], gs[i], on[i], op[i], os_[i]))
