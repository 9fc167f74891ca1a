"""Closed loop for K steps on the bench shard (mld_advance_batch between solves), then dump the inputs the population has drifted to and
the per-instance statistics of the last solve (gpurun_out/closed_loop_state.npz): the steady-state workload for CPU studies."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pyhybridcontrol_amd import gpu, host
n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]['dims']
model = gpu.GpuModel([a['mats'] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a['atoms'], d, N_p, N_t) for a in agents])
prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=800, max_pivots=40000)
prob.upload(x0, om, midx)
for k in range(steps):
    if k: prob.advance()
    st = prob.solve_resident()
    out = prob.download()
    xs, ws = prob.inputs()
    print("step %2d: %.1f ms  proven %.4f  pivots/inst %.0f nodes/inst %.1f  mean x0 %.2f min x0 %.2f  share of tanks below 50.5: %.3f" % (
        k, st['solve_ms'], (out['status'] == 0).mean(), out['pivots'].mean(), out['nodes'].mean(), xs.mean(), xs.min(), (xs < 50.5).mean()), flush=True)
tel = prob.telemetry()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'closed_loop_state.npz'),
                    x0=xs, omega=ws, midx=midx, status=out['status'], obj=out['obj'], lb=out['lower_bound'], nodes=out['nodes'], pivots=out['pivots'], latency_ns=tel['latency_ns'])
