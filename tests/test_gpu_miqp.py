"""The quadratic-cost path (north star: "batched MIQP") held to the MILP standard.

The reference states a quadratic atom as `cvx.quad_form(var, weight)` on the evolution expression of the variable
(controllers/components/objective_atoms.py:321-331 through variables.py:259-275); here K4 pulls the weight back onto the decision vector
(P = Gamma' (W + W') Gamma) and every node relaxation is a convex QP on the solver's dictionary.  Checked:

* 64 instances of the BASELINE cfg3 shape with the MIQP variant's Q_x = 1e-3 I against the C oracle at gap 1e-6;
* every returned point against an INDEPENDENT solver: binaries fixed at the returned values, the remaining convex QP solved by scipy's SLSQP on the
  ORIGINAL (un-tightened, un-scaled) rows -- objective equal to 1e-6 -- and a KKT certificate of the point by non-negative least squares;
* a cfg1-size case against full enumeration over the binaries with that independent QP solver at every leaf.
"""
import itertools

import numpy as np
import pytest
from scipy.optimize import minimize, nnls

from pyhybridcontrol_amd import gpu, host, synthetic as syn
from oracle import condense_np as cn, orc, tighten_np

pytestmark = pytest.mark.gpu


def _qp_fixed_binaries(sf, q, h, vbin, v_start):
    """min 1/2 v'Pv + q'v  s.t. G v <= h, lb <= v <= ub with the binaries fixed at vbin: scipy SLSQP on the continuous variables (independent of
    every solver in this repository); returns (objective, point)"""
    P, G = sf["cost"]["P"], sf["G"]
    isb = sf["is_bin"].astype(bool)
    c = np.where(~isb)[0]
    v0 = np.array(v_start, dtype=np.float64)
    v0[isb] = vbin
    Pcc, Pcb = P[np.ix_(c, c)], P[np.ix_(c, np.where(isb)[0])]
    qc = q[c] + Pcb @ vbin
    const = 0.5 * vbin @ P[np.ix_(np.where(isb)[0], np.where(isb)[0])] @ vbin + q[isb] @ vbin
    Gc, hc = G[:, c], h - G[:, isb] @ vbin
    rown = np.maximum(1.0, np.abs(G).max(axis=1))
    Gn, hn = Gc / rown[:, None], hc / rown
    lb, ub = sf["lb"][c], sf["ub"][c]
    bounds = [(None if not np.isfinite(a) else a, None if not np.isfinite(b) else b) for a, b in zip(lb, ub)]
    res = minimize(lambda x: 0.5 * x @ Pcc @ x + qc @ x, v0[c], jac=lambda x: Pcc @ x + qc, method="SLSQP", bounds=bounds,
                   constraints=[dict(type="ineq", fun=lambda x: hn - Gn @ x, jac=lambda x: -Gn)], options=dict(maxiter=400, ftol=1e-13))
    x = res.x
    assert np.all(Gn @ x - hn <= 1e-7), "SLSQP left the polytope"
    v = v0.copy()
    v[c] = x
    return float(res.fun + const), v


def _kkt_residual(sf, q, h, v):
    """stationarity of v over the continuous variables with multipliers >= 0 on the active rows / bounds (NNLS); relative to the gradient's size"""
    P, G = sf["cost"]["P"], sf["G"]
    c = np.where(~sf["is_bin"].astype(bool))[0]
    g = (P @ v + q)[c]
    rown = np.maximum(1.0, np.abs(G).max(axis=1))
    act = np.where((G @ v - h) / rown >= -1e-7)[0]
    at_lo = np.where(np.isfinite(sf["lb"][c]) & (v[c] - sf["lb"][c] <= 1e-7))[0]
    at_hi = np.where(np.isfinite(sf["ub"][c]) & (sf["ub"][c] - v[c] <= 1e-7))[0]
    A = np.hstack([(G[act][:, c] / rown[act, None]).T, -np.eye(len(c))[:, at_lo], np.eye(len(c))[:, at_hi]])
    if A.shape[1] == 0:
        return float(np.abs(g).max() / max(1e-12, np.abs(q).max()))
    _, rn = nnls(A, -g, maxiter=20 * A.shape[1])
    return float(rn / max(1e-12, np.linalg.norm(q[c])))


def test_miqp_cfg3_shape_64_instances_against_oracle_and_an_independent_qp_solver():
    nb = 64
    wl = syn.make_workload("cfg3", batch=nb, quadratic=True)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]),
                       gap_rel=1e-6, max_nodes=20000, max_pivots=400000)
    out = p.solve(ag["x0"], ag["omega"])
    p.close(); m.close()
    sft = cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    sf0 = cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])      # original rows: what the reference states
    isb = sf0["is_bin"].astype(bool)
    rown = np.maximum(1.0, np.abs(sf0["G"]).max(axis=1))
    proven, worst, worst_ind, worst_kkt, checked = 0, 0.0, 0.0, 0.0, 0
    for s in range(nb):
        x0, om = ag["x0"][s], ag["omega"][s]
        q, r = cn.lin_cost(sft["cost"], x0, om), cn.cost_const(sft["cost"]["const_terms"], x0, om)
        ref = orc.solve_miqp(sft["cost"]["P"], q, sft["G"], cn.rhs(sft["evo"], x0, om), sft["lb"], sft["ub"], sft["is_bin"],
                             max_nodes=20000, presolve=0, gap_rel=1e-6)
        assert ref["status"] == "optimal", (s, ref["status"])
        tot = ref["obj"] + r
        assert int(out["status"][s]) in (0, 2), (s, out["status"][s])
        v = out["v"][s]
        assert np.all((v[isb] == 0) | (v[isb] == 1)), s                                             # integer feasibility: bit-exact
        h0 = cn.rhs(sf0["evo"], x0, om)
        assert np.all((sf0["G"] @ v - h0) / rown <= 1e-6), s                                         # feasible for the ORIGINAL rows
        q0 = cn.lin_cost(sf0["cost"], x0, om)
        val = 0.5 * v @ sf0["cost"]["P"] @ v + q0 @ v + r
        assert abs(val - out["obj"][s]) <= 1e-6 * max(1.0, abs(tot)), (s, val, out["obj"][s])        # the reported objective is the point's
        assert out["lower_bound"][s] <= tot + 1e-6 * max(1.0, abs(tot)), s
        if int(out["status"][s]) == 0:
            proven += 1
            worst = max(worst, abs(out["obj"][s] - tot) / max(1.0, abs(tot)))
            assert abs(out["obj"][s] - tot) <= 2e-6 * max(1.0, abs(tot)), (s, out["obj"][s], tot)    # each within 1e-6 of the optimum
        else:
            assert out["obj"][s] >= tot - 1e-6 * max(1.0, abs(tot)), s
        if s % 4 == 0:      # independent solver on the point's binary assignment (SLSQP, original rows), and its KKT certificate
            ind, _ = _qp_fixed_binaries(sf0, q0, h0, v[isb], v)
            worst_ind = max(worst_ind, abs(ind + r - out["obj"][s]) / max(1.0, abs(tot)))
            assert abs(ind + r - out["obj"][s]) <= 1e-6 * max(1.0, abs(tot)), (s, ind + r, out["obj"][s])
            kk = _kkt_residual(sf0, q0, h0, v)
            worst_kkt = max(worst_kkt, kk)
            assert kk <= 1e-4, (s, kk)          # (the relaxations stop at a relative objective gap of 1e-10: the gradient residual is its square root)
            checked += 1
    print("MIQP cfg3 shape: proven %d of %d, worst |obj - oracle| %.2e, %d points against SLSQP: worst %.2e, KKT residual %.2e" % (proven, nb, worst, checked, worst_ind, worst_kkt))
    assert proven >= int(0.9 * nb), proven


def test_miqp_small_case_against_enumeration_with_an_independent_qp_solver():
    """cfg1 shape (one tank, N_tilde = 5: 5 binaries): all 32 assignments, each leaf QP by SLSQP; the GPU's proven optimum is the smallest"""
    wl = syn.make_workload("cfg1", batch=6, quadratic=True)
    ag = wl["agents"][0]
    d = ag["dims"]
    atoms = dict(ag["atoms"])
    atoms["Q_u"] = 0.05 * np.eye(d["nu"])
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(atoms, d, wl["N_p"], wl["N_tilde"]), max_nodes=20000)
    out = p.solve(ag["x0"], ag["omega"])
    p.close(); m.close()
    sf0 = cn.standard_form(ag["mats"], atoms, wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    isb = sf0["is_bin"].astype(bool)
    nbin = int(isb.sum())
    assert nbin <= 12
    for s in range(6):
        x0, om = ag["x0"][s], ag["omega"][s]
        q0, h0, r = cn.lin_cost(sf0["cost"], x0, om), cn.rhs(sf0["evo"], x0, om), cn.cost_const(sf0["cost"]["const_terms"], x0, om)
        best = np.inf
        for bits in itertools.product((0.0, 1.0), repeat=nbin):
            try:
                val, _ = _qp_fixed_binaries(sf0, q0, h0, np.array(bits), np.zeros(sf0["G"].shape[1]))
            except AssertionError:
                continue
            best = min(best, val + r)
        assert int(out["status"][s]) == 0 and np.isfinite(best)
        assert abs(out["obj"][s] - best) <= 1e-6 * max(1.0, abs(best)), (s, out["obj"][s], best)
