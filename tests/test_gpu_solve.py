"""GPU parity for K3 (rhs), K4 (cost) and K5/K6 (cut-and-branch MILP) through the C ABI vs the oracle."""
import numpy as np
import pytest

import condense_np as cn
import orc
import tighten_np
from pyhybridcontrol_amd import gpu, synthetic as syn, host

pytestmark = pytest.mark.gpu


def _oracle_instance(agent, wl, s, tight=True):
    d = agent["dims"]
    mats = tighten_np.tighten(agent["mats"], d, nu_l=d["nu_l"]) if tight else agent["mats"]
    sf = cn.standard_form(mats, agent["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    h = cn.rhs(sf["evo"], agent["x0"][s], agent["omega"][s])
    q = cn.lin_cost(sf["cost"], agent["x0"][s], agent["omega"][s])
    r = cn.cost_const(sf["cost"]["const_terms"], agent["x0"][s], agent["omega"][s])
    return sf, q, h, r


def check_solution(agent, wl, s, v, obj, tol=1e-6):
    """independent fp64 certificate: v is integer feasible for the ORIGINAL rows and obj is its cost"""
    sf, q, h, r = _oracle_instance(agent, wl, s, tight=False)
    G = sf["G"]
    bins = sf["is_bin"]
    assert np.all((v[bins] == 0) | (v[bins] == 1)), "binaries must be exactly 0/1"
    rown = np.maximum(1.0, np.abs(G).max(axis=1))
    assert np.all((G @ v - h) / rown <= 1e-6), "constraint violation"
    assert np.all(v >= sf["lb"] - 1e-9) and np.all(v <= sf["ub"] + 1e-9)
    assert abs(q @ v + r - obj) <= tol * max(1.0, abs(obj))


@pytest.mark.parametrize("name,nb", [("cfg1", 6), ("cfg2", 6)])
def test_solve_matches_oracle(name, nb):
    wl = syn.make_workload(name, batch=nb)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, max_nodes=20000)
    out = p.solve(ag["x0"], ag["omega"])
    for s in range(nb):
        sf, q, h, r = _oracle_instance(ag, wl, s)
        ref = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], max_nodes=20000, presolve=0)
        assert gpu._lib.STATUS_NAMES[int(out["status"][s])] == ref["status"] == "optimal", (s, out["status"][s], ref["status"])
        assert abs(out["obj"][s] - (ref["obj"] + r)) <= 1e-6 * max(1.0, abs(ref["obj"] + r)), (s, out["obj"][s], ref["obj"] + r)
        check_solution(ag, wl, s, out["v"][s], out["obj"][s])
    p.close(); m.close()
