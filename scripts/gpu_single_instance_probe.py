"""ONE solve() call to the gap on the hardest instances of a bench step: plain (one workgroup) against in-kernel hand-off settings.
   python scripts/gpu_single_instance_probe.py "100,200;50,100;50,200;200,200" [n_hard=24]"""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import bench
from pyhybridcontrol_amd import gpu, host
combos = [tuple(int(v) for v in c.split(",")) for c in (sys.argv[1] if len(sys.argv) > 1 else "100,200").split(";")]
n_hard = int(sys.argv[2]) if len(sys.argv) > 2 else 24
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 512, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=800, max_pivots=40000)
prob.upload(x0, om, midx); prob.solve_resident()
lat = prob.telemetry()["latency_ns"]
hard = np.argsort(-lat)[:n_hard]
prob.set_opts(max_nodes=20000, max_pivots=400000)
def run(tag, fn):
    ts, pr = [], 0
    for i in hard:
        t0 = time.perf_counter(); out = fn(i); ts.append((time.perf_counter() - t0) * 1e3); pr += int(out["status"][0] == 0)
    ts = np.sort(ts)
    print("%-34s p50 %7.1f ms  p90 %7.1f  max %8.1f  mean %7.1f  proven %d of %d" % (tag, ts[len(ts) // 2], ts[int(len(ts) * 0.9)], ts[-1], ts.mean(), pr, len(hard)), flush=True)
run("one workgroup (NodeLimit 20000)", lambda i: prob.solve(x0[i:i + 1], om[i:i + 1], midx[i:i + 1]))
for c in combos:
    fn_, sn_ = c[0], c[1]
    kw = dict(first_nodes=fn_, sub_nodes=sn_, max_gen=8, max_children=64, max_tree=c[2] if len(c) > 2 else 160)
    run("hand-off first %d sub %d tree %d" % (fn_, sn_, kw["max_tree"]), lambda i: prob.solve_handoff_device(x0[i:i + 1], om[i:i + 1], midx[i:i + 1], **kw))
