"""Per-instance presolve of the solve path (round 4; csrc/problem.inc s_presolve, oracle/mld_oracle.c presolve_instance).

The reference hands every instance to a MIP solver whose own presolve sees the instance's right-hand side (controllers/controller_base.py:497-512,
cvxpy -> Gurobi); the kernel's presolve is the part of that which pays here: row-activity bound propagation with the instance's x0 / omega in the
right-hand side.  Checked through the C ABI:

* with the presolve and without it (opts.reserved bit 12) the SAME optimum is proven at MIPGap 1e-6 -- it cuts off no integer-feasible point;
* it is what it claims to be: fewer dictionary rows maintained per pivot, no more pivots;
* fixings that contradict the instance are reported infeasible with and without it (with it: before the first pivot);
* the C oracle's presolve (presolve bit 2) proves the same optimum as the oracle without it, and the GPU agrees with both.
"""
import numpy as np
import pytest

from pyhybridcontrol_amd import gpu, host, synthetic as syn
from oracle import condense_np as cn, orc, tighten_np

pytestmark = pytest.mark.gpu

NO_PRESOLVE = 1 << 12


def _problem(wl, **opts):
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), **opts)
    return ag, d, m, p


@pytest.mark.parametrize("cfg,batch", [("cfg2", 64), ("cfg3", 48)])
def test_same_proven_optimum_with_and_without_presolve(cfg, batch):
    wl = syn.make_workload(cfg, batch=batch)
    res = {}
    for name, r in (("on", 0), ("off", NO_PRESOLVE)):
        ag, d, m, p = _problem(wl, gap_rel=1e-6, max_nodes=50000, max_pivots=2000000, reserved=r)
        out = p.solve(ag["x0"], ag["omega"])
        tel = p.telemetry()
        res[name] = (out, float(tel["rows_updated"].sum()), float(out["pivots"].sum()))
        p.close(); m.close()
    on, off = res["on"][0], res["off"][0]
    assert np.all(on["status"] == 0) and np.all(off["status"] == 0), (np.unique(on["status"]), np.unique(off["status"]))
    scale = np.maximum(1.0, np.abs(off["obj"]))
    assert np.all(np.abs(on["obj"] - off["obj"]) <= 2e-6 * scale), float((np.abs(on["obj"] - off["obj"]) / scale).max())
    assert np.all(on["lower_bound"] <= off["obj"] + 1e-6 * scale) and np.all(off["lower_bound"] <= on["obj"] + 1e-6 * scale)
    print("%s: rows updated %.3g -> %.3g, pivots %.0f -> %.0f" % (cfg, res["off"][1], res["on"][1], res["off"][2], res["on"][2]))
    assert res["on"][1] < 0.9 * res["off"][1]          # (measured cfg3: 0.4 x) rows that cannot bind under the instance's implied bounds are not maintained


def test_gpu_presolve_agrees_with_the_oracle_with_and_without_its_presolve():
    wl = syn.make_workload("cfg3", batch=16)
    ag, d, m, p = _problem(wl, gap_rel=1e-6, max_nodes=50000, max_pivots=2000000)
    out = p.solve(ag["x0"], ag["omega"])
    p.close(); m.close()
    sf = cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    work = {0: 0.0, 4: 0.0}
    for s in range(16):
        x0, om = ag["x0"][s], ag["omega"][s]
        q, h, r = cn.lin_cost(sf["cost"], x0, om), cn.rhs(sf["evo"], x0, om), cn.cost_const(sf["cost"]["const_terms"], x0, om)
        ref = {pre: orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], max_nodes=50000, presolve=pre, gap_rel=1e-6) for pre in (0, 4)}
        assert ref[0]["status"] == "optimal" and ref[4]["status"] == "optimal"
        tot = ref[0]["obj"] + r
        assert abs(ref[4]["obj"] - ref[0]["obj"]) <= 2e-6 * max(1.0, abs(tot)), (s, ref[4]["obj"], ref[0]["obj"])
        assert int(out["status"][s]) == 0
        assert abs(out["obj"][s] - tot) <= 2e-6 * max(1.0, abs(tot)), (s, out["obj"][s], tot)
        work[0] += ref[0]["work"]; work[4] += ref[4]["work"]
    assert work[4] < work[0], work          # (the oracle's count of row updates over the 16 instances; single instances go either way)


def test_contradicting_fixings_are_infeasible_with_and_without_presolve():
    """delta_0 = [y_0 >= 0] fixed at 0 while the load alone makes y_0 positive: no point satisfies the rows.  Every other binary stays free (255), so the
    instances go to the branch-and-cut kernel; half of the batch keeps a satisfiable fixing (delta_0 = 1) and must still be solved."""
    wl = syn.make_workload("cfg3", batch=8)
    ag0 = wl["agents"][0]
    n_h, nom = wl["n_h"], wl["n_h"] + 1
    om = ag0["omega"].copy().reshape(8, wl["N_tilde"], nom)
    om[:, 0, n_h] = 4000.0                      # load of step 0: y_0 = sum P_i u_i + 4000 > 0 whatever the heaters do
    om = om.reshape(8, -1)
    for r in (0, NO_PRESOLVE):
        ag, d, m, p = _problem(wl, gap_rel=1e-4, max_nodes=20000, reserved=r)
        nb = p.n_bin
        fixed = np.full((8, nb), 255, dtype=np.uint8)
        k_delta0 = d["nu"]                      # binaries in variable order: u_0 (nu of them, all logic inputs), then delta_0
        fixed[:4, k_delta0] = 0
        fixed[4:, k_delta0] = 1
        out = p.solve(ag["x0"], om, fixed_bin=fixed)
        p.close(); m.close()
        assert np.all(out["status"][:4] == 1), (r, out["status"])           # MLD_STATUS_INFEASIBLE
        assert np.all(out["status"][4:] == 0), (r, out["status"])
        assert np.all(out["nodes"][:4] <= 1)
        if r == 0:
            assert np.all(out["pivots"][:4] == 0), out["pivots"][:4]        # found by the propagation: not one pivot
