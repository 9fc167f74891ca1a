"""The oracle's exact MILP solver (oracle/mld_oracle.c): there is NO reference fixture for the solve (the
reference has no tests and delegates to Gurobi) -> "parity unpinned" at the solver boundary.  It is pinned here
against (a) exhaustive enumeration of the binaries, (b) scipy's HiGHS (independent third-party MILP solver),
(c) the known-answer instance recorded in SURVEY.md section 8c."""
import numpy as np
import pytest
from scipy.optimize import milp, LinearConstraint, Bounds

import condense_np as cn
import orc
import tighten_np
import bnc_np
from pyhybridcontrol_amd import synthetic as syn


def _instance(name, s=0, batch=4, tight=False, agent=0):
    wl = syn.make_workload(name, batch=batch)
    ag = wl["agents"][agent]
    d = ag["dims"]
    mats = tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]) if tight else ag["mats"]
    sf = cn.standard_form(mats, ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    h = cn.rhs(sf["evo"], ag["x0"][s], ag["omega"][s])
    q = cn.lin_cost(sf["cost"], ag["x0"][s], ag["omega"][s])
    return sf, q, h


def _highs(sf, q, h):
    lb = np.where(np.isinf(sf["lb"]), -1e7, sf["lb"])
    ub = np.where(np.isinf(sf["ub"]), 1e7, sf["ub"])
    r = milp(q, constraints=LinearConstraint(sf["G"], -np.inf, h), integrality=sf["is_bin"].astype(int),
             bounds=Bounds(lb, ub), options=dict(mip_rel_gap=0))
    assert r.status == 0
    return r.fun


def test_known_answer_dewh_survey_8c():
    """DEWH, N_tilde=5, x0=50.3, omega=[.004,.012,0,.009,.002], price [1,3,3,1,1] => u*=[1,0,0,1,0], objective 1.5"""
    mats = dict(A=[[0.9970371127900564]], B1=[[4.298192277481107]], B4=[[-179.73320827515]], b5=[[0.07407218024859108]],
                E=[[1], [-1]], F1=[[0], [0]], Psi=[[-1, 0], [0, -1]], f5=[[65.0], [-50.0]], C=[[1.0]])
    mats = {k: np.array(v, dtype=float) for k, v in mats.items()}
    price = np.array([1, 3, 3, 1, 1.0])
    atoms = dict(q_u=(price * 0.75).reshape(-1, 1), q_mu=[90.0, 90.0])
    sf = cn.standard_form(mats, atoms, 4, 5, nu_l=1)
    h = cn.rhs(sf["evo"], [50.3], [.004, .012, 0, .009, .002])
    q = cn.lin_cost(sf["cost"], [50.3], [.004, .012, 0, .009, .002])
    r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"])
    assert r["status"] == "optimal"
    u = r["x"].reshape(5, 3)[:, 0]
    assert np.array_equal(u, [1, 0, 0, 1, 0])
    assert abs(r["obj"] - 1.5) < 1e-9
    e = orc.enumerate_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"])
    assert abs(e["obj"] - 1.5) < 1e-9
    p = bnc_np.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"])
    assert p["status"] == "optimal" and abs(p["obj"] - 1.5) < 1e-9


@pytest.mark.parametrize("s", range(4))
def test_cfg1_matches_enumeration(s):
    sf, q, h = _instance("cfg1", s)
    r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"])
    e = orc.enumerate_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"])
    assert r["status"] == e["status"] == "optimal"
    assert abs(r["obj"] - e["obj"]) <= 1e-8 * max(1, abs(e["obj"]))


def test_random_small_milps_match_enumeration_and_highs():
    rng = np.random.Generator(np.random.PCG64(11))
    solved = 0
    for t in range(25):
        n, m, nb = 9, 7, 6
        G = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.7)
        xf = rng.random(n)
        h = G @ xf + rng.random(m)              # feasible for the relaxation at xf
        q = rng.standard_normal(n)
        lb = np.zeros(n)
        ub = np.concatenate([np.ones(nb), np.full(n - nb, 5.0)])
        is_bin = np.arange(n) < nb
        r = orc.solve_milp(q, G, h, lb, ub, is_bin)
        e = orc.enumerate_milp(q, G, h, lb, ub, is_bin)
        assert r["status"] == e["status"], t
        if e["status"] == "optimal":
            assert abs(r["obj"] - e["obj"]) <= 1e-7 * max(1, abs(e["obj"])), t
            hi = milp(q, constraints=LinearConstraint(G, -np.inf, h), integrality=is_bin.astype(int), bounds=Bounds(lb, ub))
            assert abs(hi.fun - e["obj"]) <= 1e-6 * max(1, abs(e["obj"]))
            xs = r["x"]
            assert np.all(G @ xs <= h + 1e-7) and np.all((xs[is_bin] == 0) | (xs[is_bin] == 1))
            solved += 1
    assert solved >= 10


@pytest.mark.parametrize("s", range(3))
def test_cfg2_matches_highs_and_tightening_keeps_the_optimum(s):
    sf0, q0, h0 = _instance("cfg2", s, tight=False)
    ref = _highs(sf0, q0, h0)
    sf, q, h = _instance("cfg2", s, tight=True)
    r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], max_nodes=50000, presolve=0)
    assert r["status"] == "optimal"
    assert abs(r["obj"] - ref) <= 1e-5 * max(1, abs(ref))
    # the returned point is feasible for the ORIGINAL (un-tightened) rows with the same cost
    x = r["x"]
    rown = np.maximum(1, np.abs(sf0["G"]).max(axis=1))
    assert np.all((sf0["G"] @ x - h0) / rown <= 1e-6)
    assert abs(q0 @ x - r["obj"]) <= 1e-9 * max(1, abs(ref))


def test_infeasible_and_node_limit_statuses():
    G = np.array([[1.0, 1.0], [-1.0, -1.0]])
    h = np.array([0.5, -1.5])                    # x1+x2 <= .5 and >= 1.5
    r = orc.solve_milp(np.ones(2), G, h, np.zeros(2), np.ones(2), np.array([True, True]))
    assert r["status"] == "infeasible" and not np.isfinite(r["obj"])
    sf, q, h = _instance("cfg2", 0, tight=False)
    r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], max_nodes=3, presolve=0)
    assert r["status"] in ("node_limit", "optimal")


def _fw_gap(P, q, G, h, lb, ub, x):
    """Frank-Wolfe optimality certificate of a convex QP, computed with HiGHS' LP: g.x - min_y g.y"""
    from scipy.optimize import linprog
    g = P @ x + q
    r = linprog(g, A_ub=G, b_ub=h, bounds=np.c_[lb, ub], method="highs")
    assert r.status == 0
    return float(g @ x - r.fun)


def test_convex_qp_relaxation_is_optimal_by_certificate():
    rng = np.random.Generator(np.random.PCG64(3))
    for t in range(8):
        n, m, r = 12, 9, 5
        G = rng.standard_normal((m, n))
        xf = rng.random(n)
        h = G @ xf + rng.random(m)
        R = rng.standard_normal((r, n))
        P, q = R.T @ R, rng.standard_normal(n)
        lb, ub = np.zeros(n), np.full(n, 3.0)
        res = orc.solve_miqp(P, q, G, h, lb, ub, np.zeros(n, bool))
        assert res["status"] == "optimal"
        x = res["x"]
        assert np.all(G @ x <= h + 1e-8) and np.all(x >= -1e-9) and np.all(x <= 3 + 1e-9)
        assert abs(0.5 * x @ P @ x + q @ x - res["obj"]) < 1e-9
        assert _fw_gap(P, q, G, h, lb, ub, x) < 1e-7


def test_small_miqp_matches_enumeration_of_the_binaries():
    import itertools
    rng = np.random.Generator(np.random.PCG64(4))
    for t in range(6):
        n, m, nb, r = 10, 7, 5, 4
        G = rng.standard_normal((m, n))
        xf = rng.random(n)
        h = G @ xf + rng.random(m)
        R = rng.standard_normal((r, n))
        P, q = R.T @ R, rng.standard_normal(n)
        lb = np.zeros(n)
        ub = np.concatenate([np.ones(nb), np.full(n - nb, 3.0)])
        isb = np.arange(n) < nb
        res = orc.solve_miqp(P, q, G, h, lb, ub, isb)
        best = np.inf
        for bits in itertools.product([0, 1], repeat=nb):
            l, u = lb.copy(), ub.copy()
            l[:nb] = u[:nb] = bits
            rr = orc.solve_miqp(P, q, G, h, l, u, np.zeros(n, bool))
            if rr["status"] == "optimal":
                assert _fw_gap(P, q, G, h, l, u, rr["x"]) < 1e-7
                best = min(best, rr["obj"])
        assert res["status"] == ("optimal" if np.isfinite(best) else "infeasible")
        if np.isfinite(best):
            assert abs(res["obj"] - best) <= 1e-7 * max(1, abs(best))


def test_miqp_with_zero_hessian_equals_milp():
    sf, q, h = _instance("cfg2", 1, tight=True)
    a = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], presolve=0)
    b = orc.solve_miqp(np.zeros((q.size, q.size)), q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], presolve=0)
    assert a["status"] == b["status"] == "optimal" and abs(a["obj"] - b["obj"]) <= 1e-8 * max(1, abs(a["obj"]))


def test_unbounded_problem_is_reported_not_boxed():
    """a free variable with a cost and no bound in its direction: status 'unbounded', objective -inf (the solver boxes free
    variables at +-1e7 internally; resting on that box must not be reported as an optimum)"""
    r = orc.solve_milp(np.array([1.0, 0.0]), np.array([[0.0, 1.0]]), np.array([1.0]), np.array([-np.inf, 0.0]),
                       np.array([np.inf, 1.0]), np.array([0, 1], np.uint8))
    assert r["status"] == "unbounded" and r["obj"] == -np.inf
    r = orc.solve_milp(np.array([1.0, 0.0]), np.array([[-1.0, 1.0]]), np.array([1.0]), np.array([-np.inf, 0.0]),
                       np.array([np.inf, 1.0]), np.array([0, 1], np.uint8))       # x >= y - 1 bounds it: optimum -1 at y = 0
    assert r["status"] == "optimal" and abs(r["obj"] + 1.0) < 1e-9


def test_oracle_against_highs_golden_on_bench_instances():
    """the C oracle (checker of the GPU tests, CPU baseline of bench.py) against optima nobody here computed: the first 32
    instances of the bench shard vs tests/golden/solve_cfg4_bench.npz (scipy HiGHS, gap 0, original rows)"""
    import os
    import bench
    import tighten_np
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solve_cfg4_bench.npz"))
    agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 1, 0)
    n = 32
    qs, Gs, hs, rc = [], [], [], []
    for i in range(n):
        ag = agents[int(midx[i])]
        d = ag["dims"]
        sf = cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], N_p, N_t, nu_l=d["nu_l"])
        qs.append(cn.lin_cost(sf["cost"], x0[i], om[i])); hs.append(cn.rhs(sf["evo"], x0[i], om[i])); Gs.append(sf["G"])
        rc.append(cn.cost_const(sf["cost"]["const_terms"], x0[i], om[i]))
    opt = gold["obj"][:n]
    scale = np.maximum(1e-9, np.abs(opt))
    for gap, nodes in ((1e-2, 800), (1e-6, 20000)):
        r, _ = orc.solve_milp_batch(qs, Gs, hs, sf["lb"], sf["ub"], sf["is_bin"], threads=4, gap_rel=gap, max_nodes=nodes, presolve=0, max_pivots=400000)
        obj, lb = r["obj"] + np.array(rc), r["lower_bound"] + np.array(rc)
        assert np.all(obj >= opt - 1e-6 * scale) and np.all(lb <= opt + 1e-6 * scale)
        proven = r["status"] == 0
        assert proven.sum() >= n - 2
        assert np.all(obj[proven] - opt[proven] <= gap * np.abs(obj[proven]) + 1e-6 * scale[proven])


# ---- round 3: MIP start, anti-stalling perturbation, new fixtures -----------------------------------------------

def _bench_instance(i):
    import bench
    agents, N_p, N_t, x0, om, midx = bench.make_shard(64, i // 64 + 1, 0)
    ag = agents[int(midx[i])]
    d = ag["dims"]
    sf = cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], N_p, N_t, nu_l=d["nu_l"])
    return sf, cn.lin_cost(sf["cost"], x0[i], om[i]), cn.rhs(sf["evo"], x0[i], om[i]), cn.cost_const(sf["cost"]["const_terms"], x0[i], om[i])


def test_mip_start_never_changes_the_proven_optimum():
    """orc_solve_miqp_start: the start is only ever an incumbent candidate (the reference forwards warm_start=True to its backend,
    controller_base.py:493,509-512) -- the optimum, a wrong start, an infeasible-looking start all end at the same proven value"""
    sf, q, h = _instance("cfg2", 1, tight=True)
    kw = dict(gap_rel=0.0, max_nodes=100000, presolve=0)
    ref = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], **kw)
    assert ref["status"] == "optimal"
    rng = np.random.default_rng(3)
    for start in (ref["x"], np.zeros_like(ref["x"]), np.ones_like(ref["x"]), rng.integers(0, 2, ref["x"].size).astype(float)):
        r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], x_start=start, **kw)
        assert r["status"] == "optimal" and abs(r["obj"] - ref["obj"]) <= 1e-9 * max(1.0, abs(ref["obj"]))


def test_mip_start_is_evaluated_first():
    """round 4: a MIP start is evaluated FIRST (its leaf LP from the slack basis, the root relaxation from there), and the cut loop stops once the bound is within the gap of it.  A poor start
    (all binaries 0) costs that leaf and changes nothing else about the answer; the optimal point as start ends the solve at the root with fewer pivots
    than the cold solve of an instance that needs a tree"""
    sf, q, h, _ = _bench_instance(98)              # proven at the root
    kw = dict(gap_rel=1e-2, max_nodes=800, presolve=0, max_pivots=40000)
    a = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], **kw)
    b = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], x_start=np.zeros(q.size), **kw)
    assert a["nodes"] <= 2 and a["status"] == b["status"] == "optimal"
    assert abs(b["obj"] - a["obj"]) <= 1e-2 * abs(a["obj"]) + 1e-9 and b["lower_bound"] <= a["obj"] + 1e-9
    sf, q, h, _ = _bench_instance(1907)            # needs cuts and a search when solved cold
    cold = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], **kw)
    warm = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], x_start=cold["x"], **kw)
    assert warm["status"] == "optimal" and warm["obj"] <= cold["obj"] + 1e-9 * abs(cold["obj"])
    assert warm["nodes"] <= cold["nodes"] + 1


def test_cost_perturbation_breaks_the_stall_and_keeps_the_value():
    """bench instance 1907 (PV surplus in 20 of 25 steps: dual degenerate): without the perturbation one cut round stalls for
    > 2000 Bland pivots and the root is rebuilt without cuts; with it the same optimum costs a fraction of the pivots"""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path[:0] = [%r, %r, %r]; import test_oracle_solver as t, orc\n"
            "sf, q, h, r = t._bench_instance(1907)\n"
            "o = orc.solve_milp(q, sf['G'], h, sf['lb'], sf['ub'], sf['is_bin'], gap_rel=1e-2, max_nodes=800, presolve=0, max_pivots=40000)\n"
            "print(o['status'], repr(o['obj'] + r), o['pivots'], int(o['bland']), repr(o['lower_bound'] + r))\n"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"),
               os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
    res = {}
    for tag, env in (("on", {}), ("off", {"ORC_NO_PERT": "1"})):      # the switch is read once per process
        out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **env)).decode().split()
        res[tag] = dict(status=out[0], obj=float(out[1]), pivots=int(out[2]), bland=int(out[3]), lb=float(out[4]))
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "solve_cfg4_bench.npz"))["obj"][1907]
    for tag in ("on", "off"):
        assert res[tag]["obj"] >= gold - 1e-6 * abs(gold) and res[tag]["lb"] <= gold + 1e-6 * abs(gold), (tag, res[tag], gold)
        if res[tag]["status"] == "optimal":
            assert res[tag]["obj"] - gold <= 1e-2 * abs(res[tag]["obj"]) + 1e-9
    assert res["off"]["bland"] > 1000 and res["on"]["bland"] < 0.2 * res["off"]["bland"], res
    assert res["on"]["pivots"] < 0.6 * res["off"]["pivots"], res


def test_round3_fixtures_are_consistent():
    """the committed HiGHS optima of bench.py's timed scenario set and of the steady-state closed-loop inputs: proven, bracketed, and
    the oracle -- a different algorithm -- agrees on a sample at the bench's options"""
    import os
    import bench
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    timed = np.load(os.path.join(gdir, "solve_cfg4_timed.npz"))
    cl = np.load(os.path.join(gdir, "solve_cfg4_closed_loop.npz"))
    z = np.load(os.path.join(gdir, "closed_loop_cfg4_inputs.npz"))
    assert timed["obj"].size == 1024 and np.all(timed["proven"] == 1) and np.all(timed["dual_bound"] <= timed["obj"] + 1e-6 * np.abs(timed["obj"]) + 1e-9)
    assert cl["obj"].size == z["x0"].shape[0] == 256 and cl["proven"].mean() >= 0.95
    assert z["x0"].min() > 45.0 and z["x0"].mean() < 58.5, "steady state sits near the lower temperature bound, below the seeded 55..64"
    agents, N_p, N_t, _, _, _ = bench.make_shard(64, 1, 0)
    xs, ws = bench.step_scenarios(0, 1, 64 * 512)
    forms = {}
    for kind, idx in (("timed", (0, 65, 130, 1023)), ("cl", (3, 100, 255))):
        for i in idx:
            a = int(i % 64) if kind == "timed" else int(z["model_idx"][i])
            if a not in forms:
                d = agents[a]["dims"]
                forms[a] = cn.standard_form(tighten_np.tighten(agents[a]["mats"], d, nu_l=d["nu_l"]), agents[a]["atoms"], N_p, N_t, nu_l=d["nu_l"])
            sf = forms[a]
            x0, om = (xs[i], ws[i]) if kind == "timed" else (z["x0"][i], z["omega"][i])
            h, q = cn.rhs(sf["evo"], x0, om), cn.lin_cost(sf["cost"], x0, om)
            rc = cn.cost_const(sf["cost"]["const_terms"], x0, om)
            r = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], gap_rel=1e-2, max_nodes=800, presolve=0, max_pivots=40000)
            opt = (timed if kind == "timed" else cl)["obj"][i]
            assert r["obj"] + rc >= opt - 1e-6 * abs(opt) and r["lower_bound"] + rc <= opt + 1e-6 * abs(opt), (kind, i)
            if r["status"] == "optimal":
                assert r["obj"] + rc - opt <= 1e-2 * abs(r["obj"] + rc) + 1e-9, (kind, i)


def test_instance_presolve_keeps_the_optimum_and_pays():
    """round 4 (presolve bit 2; csrc/problem.inc s_presolve runs the same passes): row-activity bound propagation on the instance's own right-hand side.  It cuts
    off no integer-feasible point -- the proven optimum at gap 0 is the one without it and HiGHS' --, and on the cfg3 shape it halves the row updates"""
    work = {0: 0.0, 4: 0.0}
    for s in range(4):
        sf, q, h = _instance("cfg3", s, batch=4, tight=True)
        res = {pre: orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], gap_rel=0.0, max_nodes=100000, presolve=pre) for pre in (0, 4)}
        assert res[0]["status"] == res[4]["status"] == "optimal"
        assert abs(res[4]["obj"] - res[0]["obj"]) <= 1e-7 * max(1.0, abs(res[0]["obj"])), (s, res[4]["obj"], res[0]["obj"])
        if s == 0:
            ref = _highs(sf, q, h)
            assert abs(res[4]["obj"] - ref) <= 1e-6 * max(1.0, abs(ref))
        # the point the presolved search returns satisfies the ORIGINAL rows and bounds
        x = res[4]["x"]
        assert np.all(sf["G"] @ x - h <= 1e-6 * np.maximum(1.0, np.abs(h)))
        assert np.all(x >= sf["lb"] - 1e-9) and np.all(x <= sf["ub"] + 1e-9)
        work[0] += res[0]["work"]; work[4] += res[4]["work"]
    assert work[4] < 0.8 * work[0], work


def test_instance_presolve_detects_contradicting_fixings_without_a_pivot():
    """delta_0 = [y_0 >= 0] fixed at 0 while the load alone makes y_0 positive (tests/test_gpu_presolve.py runs the same case on the GPU)"""
    wl = syn.make_workload("cfg3", batch=1)
    ag = wl["agents"][0]
    d = ag["dims"]
    sf = cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    om = ag["omega"][0].copy().reshape(wl["N_tilde"], d["nomega"])
    om[0, d["nomega"] - 1] = 4000.0
    om = om.ravel()
    h, q = cn.rhs(sf["evo"], ag["x0"][0], om), cn.lin_cost(sf["cost"], ag["x0"][0], om)
    lb, ub = sf["lb"].copy(), sf["ub"].copy()
    j_delta0 = d["nu"]                          # step 0: u (nu), delta (1), z, mu
    assert sf["is_bin"][j_delta0]
    lb[j_delta0] = ub[j_delta0] = 0.0
    with_pre = orc.solve_milp(q, sf["G"], h, lb, ub, sf["is_bin"], max_nodes=1000, presolve=4)
    without = orc.solve_milp(q, sf["G"], h, lb, ub, sf["is_bin"], max_nodes=1000, presolve=0)
    assert with_pre["status"] == without["status"] == "infeasible"
    assert with_pre["pivots"] == 0 and without["pivots"] > 0
    lb[j_delta0] = ub[j_delta0] = 1.0           # the satisfiable fixing is still solved
    ok = orc.solve_milp(q, sf["G"], h, lb, ub, sf["is_bin"], max_nodes=20000, presolve=4, gap_rel=1e-4)
    assert ok["status"] == "optimal"
