"""Branch-and-cut on the LDS-resident formulation (k_milp_lds, opts.reserved bit 10) against the dense-dictionary kernel on the same
instances:  python scripts/gpu_milp_lds_probe.py [cfg1|cfg2|cfg3|cfg4] [batch] [max_nodes] [gap_rel]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyhybridcontrol_amd import gpu, host, synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
max_nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
gap_rel = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
extra = int(os.environ.get("ML_FLAGS", "0"))
kw_all = dict(cut_rounds=int(os.environ["ML_CUT_ROUNDS"])) if "ML_CUT_ROUNDS" in os.environ else {}
kw_dense = dict(mir_per_round=int(os.environ["ML_DENSE_MIR"])) if "ML_DENSE_MIR" in os.environ else {}
wl = syn.make_workload(name, batch=nb); ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
res = {}
for tag, flags in (("dense", 0), ("k_milp_lds", 1024 | extra)):
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, reserved=flags, max_nodes=max_nodes, gap_rel=gap_rel, gap_abs=1e-9 if gap_rel == 0 else 0.0, **kw_all, **(kw_dense if tag == "dense" else {}))
    p.upload(ag["x0"], ag["omega"])
    t0 = time.time(); st = p.solve_resident(); wall = time.time() - t0
    out = p.download(); tel = p.telemetry()
    res[tag] = out
    print("%-10s %s batch %d: rhs+solve %.1f ms (wall %.1f)  %.0f solves/s  pivots %.1f nodes %.1f cuts %.1f  us/pivot/wg %.2f  status: opt %d inf %d nodelim %d other %d" % (
        tag, name, nb, st["solve_ms"], wall * 1e3, nb / wall, out["pivots"].mean(), out["nodes"].mean(), st["cuts"] / nb,
        tel["latency_ns"].sum() * 1e-3 / max(1, out["pivots"].sum()),
        (out["status"] == 0).sum(), (out["status"] == 1).sum(), (out["status"] == 2).sum(), ((out["status"] != 0) & (out["status"] != 1) & (out["status"] != 2)).sum()), flush=True)
    if tag == "k_milp_lds":
        import ctypes as C
        from pyhybridcontrol_amd import _lib
        prof = (C.c_int64 * 8)(); _lib.load().mld_debug_profile(p._h, prof)
        t = np.array(list(prof), dtype=float); names = ["simplex", "refresh/verify", "penalties", "bound changes", "leaf", "cut derivation", "bookkeeping", "set-up"]
        if extra & 4096:
            names = ["leaving", "rho", "pivot row", "ratio test", "entering column", "value updates", "W^-1 update", "outside pivots"]
            print("   simplex phases (us per pivot): " + ", ".join("%s %.2f" % (nm, v / 100.0 / out["pivots"].sum()) for nm, v in zip(names, t)))
        else: print("   phases (us per node): " + ", ".join("%s %.1f" % (nm, v / 100.0 / out["nodes"].sum()) for nm, v in zip(names, t)) + "; pivots per node %.2f" % (out["pivots"].sum() / out["nodes"].sum()))
        print("   status -1 (would fall back to the dense kernel; only visible with ML_FLAGS=2048): %d" % (out["status"] == -1).sum())
    p.close()
a, b = res["dense"], res["k_milp_lds"]
both = (a["status"] == 0) & (b["status"] == 0)
rel = np.abs(a["obj"][both] - b["obj"][both]) / np.maximum(1.0, np.abs(a["obj"][both]))
print("both optimal %d / %d; max rel objective difference %.3e; status equal %d / %d" % (both.sum(), nb, rel.max() if both.any() else float("nan"), (a["status"] == b["status"]).sum(), nb))
bad = np.where(both)[0][rel > 1e-6] if both.any() else []
for i in list(bad)[:10]: print("   inst %d dense %.9g lds %.9g (lb %.9g / %.9g)" % (i, a["obj"][i], b["obj"][i], a["lower_bound"][i], b["lower_bound"][i]))
