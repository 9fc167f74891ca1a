"""Tight-gap optimality claims of the C oracle (same algorithm as k_solve) against HiGHS at mip_rel_gap 1e-7 on cfg3 instances,
under a deep root cut loop:  python scripts/cpu_cut_validity.py [batch=96] [oracle.so] [cut_rounds=12 cuts_per_round=80 max_cuts=400 ...]
(round 1 recorded one false "optimal" -- 23.1755 against 23.1641 -- at 12 rounds x 80 Gomory cuts in this batch)."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
_G = {}


def _init(nb, lib, kw):
    import condense_np as cn
    import orc
    import tighten_np
    from pyhybridcontrol_amd import synthetic as syn
    if lib:
        orc._lib = C.CDLL(lib)
        orc._lib.orc_solve_milp.restype = C.c_int
    wl = syn.make_workload("cfg3", batch=nb)
    ag = wl["agents"][0]
    d = ag["dims"]
    _G.update(ag=ag, kw=kw, sf=cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"]),
              raw=cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"]))


def _one(s):
    import condense_np as cn
    import orc
    from scipy.optimize import Bounds, LinearConstraint, milp
    ag, sf, raw = _G["ag"], _G["sf"], _G["raw"]
    x0, om = ag["x0"][s], ag["omega"][s]
    r = orc.solve_milp(cn.lin_cost(sf["cost"], x0, om), sf["G"], cn.rhs(sf["evo"], x0, om), sf["lb"], sf["ub"], sf["is_bin"],
                       gap_rel=1e-4, max_nodes=2000, presolve=0, **_G["kw"])
    if r["status"] != "optimal":
        return s, r["status"], r["obj"], np.nan
    ref = milp(cn.lin_cost(raw["cost"], x0, om), constraints=LinearConstraint(raw["G"], -np.inf, cn.rhs(raw["evo"], x0, om)),
               integrality=raw["is_bin"].astype(int), bounds=Bounds(raw["lb"], raw["ub"]), options=dict(mip_rel_gap=1e-7, time_limit=60))
    return s, r["status"], r["obj"], (ref.fun if ref.status == 0 else np.nan)


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    lib = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].endswith(".so") else None
    kw = dict(cut_rounds=12, cuts_per_round=80, max_cuts=400, mir_per_round=20)
    for a in sys.argv[2:]:
        if "=" in a:
            k, v = a.split("=")
            kw[k] = int(v)
    with mp.Pool(8, initializer=_init, initargs=(nb, lib, kw)) as pool:
        res = pool.map(_one, range(nb), chunksize=2)
    bad = [(s, o, h, (o - h) / max(1.0, abs(h))) for s, st, o, h in res if st == "optimal" and np.isfinite(h) and ((o - h) / max(1.0, abs(h)) > 2e-4 or (o - h) / max(1.0, abs(h)) < -1e-6)]
    print("options", kw, "lib", lib or "current", "| proven", sum(st == "optimal" for _, st, _, _ in res), "of", nb, "| claims off by > 2e-4 (or below the optimum):", bad)
