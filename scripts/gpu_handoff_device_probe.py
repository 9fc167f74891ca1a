"""in-kernel hand-off (mld_set_handoff) on the bench shard: wall time and proven share over (first_nodes, sub_nodes, max_gen, max_children)
   python scripts/gpu_handoff_device_probe.py [gap=1e-6] [n_scen=512] "300,200,10,64;150,200,10,64" """
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import bench
from pyhybridcontrol_amd import gpu, host
gap = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-6
n_scen = int(sys.argv[2]) if len(sys.argv) > 2 else 512
combos = [tuple(int(x) for x in c.split(",")) for c in (sys.argv[3] if len(sys.argv) > 3 else "300,200,10,64").split(";")]
combos = [c if len(c) > 4 else c + (160,) for c in combos]
combos = [c if len(c) > 6 else c + (0, 0) for c in combos]
agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
d = agents[0]["dims"]
model = gpu.GpuModel([a["mats"] for a in agents], d)
cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
exact = gap < 1e-3
prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=gap, max_nodes=20000 if exact else 800, max_pivots=400000 if exact else 40000)
gold = np.load("tests/golden/solve_cfg4_bench.npz")["obj"]
n = x0.shape[0]
for (fn, sn, mg, mc, mt, dn, rn) in combos:
    # learn the queue order on another scenario set first, as the bench does
    xs, ws = bench.step_scenarios(0, 1, n)
    prob.set_handoff(True, sub_nodes=sn, max_gen=mg, max_children=mc, max_tree=mt, room_factor=4.0, donate=dn, rounds=rn)
    prob.set_opts(max_nodes=fn, gap_rel=1e-2, max_pivots=40000)
    prob.upload(xs, ws, midx); prob.solve_resident()
    prob.set_opts(max_nodes=fn, gap_rel=gap, max_pivots=400000 if exact else 40000)
    t0 = time.perf_counter()
    prob.upload(x0, om, midx)
    st = prob.solve_resident()
    out = prob.download()
    wall = time.perf_counter() - t0
    hs = prob.handoff_stats()
    k = min(n, gold.size)
    rel = (out["obj"][:k] - gold[:k]) / np.maximum(1e-9, np.abs(gold[:k]))
    lim = out["status"] == 2
    with np.errstate(invalid="ignore"):
        g = (out["obj"] - out["lower_bound"]) / np.maximum(1e-9, np.abs(out["obj"]))
    print("first %d sub %d gen %d children %d tree %d donate %d x %d: %.0f /s  wall %.2f s kernel %.2f s  proven %.5f  unfinished %d items %d  worst rel above optimum %.2e (checked %d) below %d  gap of limited max %.3f  nodes/inst %.1f pivots/inst %.1f" % (
        fn, sn, mg, mc, mt, dn, rn, n / wall, wall, st["solve_ms"] * 1e-3, (out["status"] == 0).mean(), int(lim.sum()), hs["items"] * 1000000 + hs["given_up"] * 1000 + hs["queue_full"], rel.max(), k, int((rel < -1e-6).sum()),
        float(np.nanmax(g[lim])) if lim.any() else 0.0, out["nodes"].mean(), out["pivots"].mean()), flush=True)
    if os.environ.get("HO_ENTRIES"):
        import ctypes as C
        from pyhybridcontrol_amd import _lib
        cap = 10 * n
        tk, gn, stt, rt = np.zeros(cap, np.int64), np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        lib = _lib.load(); lib.mld_debug_entries.restype = C.c_int
        k = lib.mld_debug_entries(prob._h, cap, tk.ctypes.data_as(C.POINTER(C.c_int64)), gn.ctypes.data_as(C.POINTER(C.c_int32)), stt.ctypes.data_as(C.POINTER(C.c_int32)), rt.ctypes.data_as(C.POINTER(C.c_int32)))
        tk, gn, stt, rt = tk[:k] * 1e-5, gn[:k], stt[:k], rt[:k]      # ms (100 MHz clock)
        print("   entries %d busy %.2f s per workgroup (sum of in-kernel times / 256) of %.2f s kernel" % (k, tk.sum() / 256 * 1e-3, st["solve_ms"] * 1e-3))
        for g in range(0, gn.max() + 1):
            mk = gn == g
            print("   gen %2d: %6d entries  busy %.3f s/256  mean %.1f ms max %.0f ms  status counts %s" % (g, mk.sum(), tk[mk].sum() / 256e3, tk[mk].mean(), tk[mk].max(), dict(zip(*np.unique(stt[mk], return_counts=True)))))
prob.set_handoff(False)
