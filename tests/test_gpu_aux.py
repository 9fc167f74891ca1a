"""GPU parity for the per-step auxiliary resolution: MldModel.lsim_k / _compute_aux
(models/mld_model.py:647-766) as a horizon-1 instance of the batched path, vs the C oracle and vs the reference's
own constraint expression."""
import numpy as np
import pytest

import condense_np as cn
import orc
import pyhybridcontrol_amd as phc
from pyhybridcontrol_amd import synthetic as syn
from pyhybridcontrol_amd.aux_resolve import AuxResolver, fold_known

pytestmark = pytest.mark.gpu


def _triples(ag, B, seed):
    rng = np.random.default_rng(seed)
    d = ag["dims"]
    x = rng.uniform(48.0, 66.0, size=(B, d["nx"]))
    u = (rng.uniform(size=(B, d["nu"])) < 0.4).astype(float)
    om = ag["omega"][rng.integers(0, ag["omega"].shape[0], B), :d["nomega"]]
    return x, u, om


def _residual(mats, d, x, u, om, dl, z, mu):
    g = lambda k, r, c: np.zeros((r, c)) if mats.get(k) is None or np.size(mats[k]) == 0 else np.asarray(mats[k], float).reshape(r, c)
    nx, ny, nc = d["nx"], d["ny"], d["nc"]
    y = g("C", ny, nx) @ x + g("D1", ny, d["nu"]) @ u + g("D2", ny, d["ndelta"]) @ dl + g("D3", ny, d["nz"]) @ z + \
        g("D4", ny, d["nomega"]) @ om + g("d5", ny, 1)[:, 0]
    return (g("E", nc, nx) @ x + g("F1", nc, d["nu"]) @ u + g("F2", nc, d["ndelta"]) @ dl + g("F3", nc, d["nz"]) @ z +
            g("F4", nc, d["nomega"]) @ om + g("G", nc, ny) @ y + g("Psi", nc, d["nmu"]) @ mu - g("f5", nc, 1)[:, 0])


@pytest.mark.parametrize("name", ["cfg2", "cfg3"])
def test_aux_resolution_matches_oracle_and_reference_constraints(name):
    wl = syn.make_workload(name, batch=8)
    ag = wl["agents"][0]
    d = ag["dims"]
    B = 48
    x, u, om = _triples(ag, B, 7)
    res = AuxResolver(ag["mats"], d)
    out = res.resolve(x, u, om)
    assert np.all(out["status"] == 0)
    m2, d2, known = fold_known(ag["mats"], d, ("delta", "z", "mu"))
    sf = cn.standard_form(m2, {"q_mu": np.ones((d2["nmu"], 1))}, 0, 1, nu_l=0)
    for s in range(B):
        w2 = np.concatenate([om[s], u[s]])
        h = cn.rhs(sf["evo"], x[s], w2)
        q = cn.lin_cost(sf["cost"], x[s], w2)
        ref = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], max_nodes=20000, presolve=0)
        assert ref["status"] == "optimal"
        mu_sum = out["mu"][s].sum()
        assert abs(mu_sum - ref["obj"]) <= 1e-6 * max(1.0, abs(ref["obj"])), (s, mu_sum, ref["obj"])
        assert np.all((out["delta"][s] == 0) | (out["delta"][s] == 1))
        assert np.all(out["mu"][s] >= -1e-9)
        # the reference's own feasibility statement (mld_model.py:735-744) holds for the returned point
        r = _residual(ag["mats"], d, x[s], u[s], om[s], out["delta"][s], out["z"][s], out["mu"][s])
        assert r.max() <= 1e-6 * max(1.0, np.abs(x[s]).max()), (s, r.max())
    res.close()


def test_lsim_k_resolves_missing_auxiliaries_on_the_gpu():
    """lsim_k(x, u, omega) with delta/z/mu left at ParNotSet == lsim_k with the resolver's values given; a known
    delta is honoured (folded into the disturbance channel) and None still means zeros."""
    wl = syn.make_workload("cfg2", batch=4)
    ag = wl["agents"][0]
    d = ag["dims"]
    model = phc.MldModel(ag["mats"], nu_l=d["nu_l"])
    x, u, om = _triples(ag, 3, 11)
    for s in range(3):
        a = model.lsim_k(x_k=x[s], u_k=u[s], omega_k=om[s])
        assert a["delta"].shape == (d["ndelta"], 1) and a["mu"].shape == (d["nmu"], 1)
        b = model.lsim_k(x_k=x[s], u_k=u[s], omega_k=om[s], delta_k=a["delta"], z_k=a["z"], mu_k=a["mu"])
        assert np.array_equal(a["x_k1"], b["x_k1"]) and np.array_equal(a["y"], b["y"])
        c = model.lsim_k(x_k=x[s], u_k=u[s], omega_k=om[s], delta_k=a["delta"])          # z, mu resolved for that delta
        assert np.allclose(c["z"], a["z"], atol=1e-7) and abs(c["mu"].sum() - a["mu"].sum()) <= 1e-6
        r = _residual(ag["mats"], d, x[s], u[s], om[s], a["delta"][:, 0], a["z"][:, 0], a["mu"][:, 0])
        assert r.max() <= 1e-6 * np.abs(x[s]).max()
    with pytest.raises(ValueError):
        model.lsim_k(x_k=x[0], u_k=u[0])                       # omega_k is required, as in the reference
    with pytest.raises(ValueError):
        model.lsim_k(x_k=x[0], v_k=np.zeros(model.mld_info.nv), u_k=u[0], omega_k=om[0])
