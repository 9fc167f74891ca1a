"""Relaxation-only mode (every binary fixed: one LP per instance) -- the LDS-resident revised simplex (k_lp_lds) against the dense-dictionary
kernel on the same instances:  python scripts/gpu_lp_probe.py [cfg2|cfg3] [batch]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pyhybridcontrol_amd import gpu, host, synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
wl = syn.make_workload(name, batch=nb); ag = wl["agents"][0]; d = ag["dims"]
m = gpu.GpuModel([ag["mats"]], d)
cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
rng = np.random.Generator(np.random.PCG64(9))
for tag, kw in (("k_lp_lds", dict()), ("dense", dict(reserved=256))):
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, **kw)
    if tag == "k_lp_lds":
        bins = np.where(p.is_bin)[0]; nv = d["nu"] + d["ndelta"] + d["nz"] + d["nmu"]; isdelta = (bins % nv) == d["nu"]
        fixed = np.zeros((nb, p.n_bin), dtype=np.uint8)
        om = ag["omega"].reshape(nb, wl["N_tilde"], -1)
        u = (rng.random((nb, wl["N_tilde"], d["nu"])) < 0.2).astype(np.uint8)
        y = u @ ag["params"]["P_h_Nom"] + om[:, :, -1]
        fixed[:, ~isdelta] = u.reshape(nb, -1); fixed[:, isdelta] = (y >= 0)
    p.upload(ag["x0"], ag["omega"], fixed_bin=fixed); p.solve_resident(); st = p.solve_resident(); out = p.download(); tel = p.telemetry()
    print("%-9s %s batch %d: %.3f ms  %.0f LPs/s  pivots/LP %.1f  us per pivot per workgroup %.2f  optimal %d infeasible %d other %d" % (
        tag, name, nb, st["solve_ms"], nb / st["solve_ms"] * 1e3, out["pivots"].mean(), tel["latency_ns"].sum() * 1e-3 / max(1, out["pivots"].sum()),
        (out["status"] == 0).sum(), (out["status"] == 1).sum(), ((out["status"] != 0) & (out["status"] != 1)).sum()), flush=True)
    if tag == "k_lp_lds":
        ref = out
        import ctypes as C
        from pyhybridcontrol_amd import _lib
        prof = (C.c_int64 * 8)(); _lib.load().mld_debug_profile(p._h, prof)
        t = np.array(list(prof), dtype=float); names = ["leaving", "rho", "pivot row", "ratio test", "entering column", "value updates", "W^-1 update", "set-up/refresh/verify"]
        print("   phases (us per pivot): " + ", ".join("%s %.2f" % (nm, v / 100.0 / out["pivots"].sum()) for nm, v in zip(names, t)))
    else: print("max |obj diff| rel %.2e, status equal %s" % (np.nanmax(np.abs(ref["obj"] - out["obj"])[np.isfinite(out["obj"])] / np.maximum(1, np.abs(out["obj"][np.isfinite(out["obj"])]))), np.array_equal(ref["status"], out["status"])))
    p.close()
