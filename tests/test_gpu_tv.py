"""GPU parity for time-varying horizons: one MLD model per horizon step, condensed by k_condense_tv and
solved by the unchanged downstream kernels, through the C ABI vs the oracle (condense_np.condense_tv + mld_oracle.c)."""
import os

import numpy as np
import pytest

import _golden as g
import _tv
import condense_np as cn
import orc
import tighten_np
from pyhybridcontrol_amd import gpu, synthetic as syn, host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", g.case_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_identical_steps_reproduce_reference_golden(path):
    z, mats, dims, N_p, N_t = g.load_case(path)
    m = gpu.GpuModel([[mats] * N_t], dims, time_varying=True)
    evo = m.condense(N_t)
    for name in g.EVO_NAMES:
        g.check_evo(z, name, evo[name][0], dims)
    m.close()


@pytest.mark.parametrize("chain_only", [False, True], ids=["wide", "chain"])
@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4"])
def test_condense_tv_matches_oracle(name, chain_only, monkeypatch):
    """both device paths: k_tv_chain + k_tv_rows (shapes that fit LDS) and the single-kernel k_condense_tv kept for the rest"""
    if chain_only:
        monkeypatch.setenv("MLD_TV_CHAIN_ONLY", "1")
    wl = syn.make_workload(name, batch=1, n_agents=3)
    N, dims = wl["N_tilde"], wl["agents"][0]["dims"]
    horizons = [_tv.step_models(a["mats"], N, seed=11 + i, strength=0.2) for i, a in enumerate(wl["agents"])]
    m = gpu.GpuModel(horizons, dims, time_varying=True)
    assert m.n_models == 3
    evo = m.condense(N)
    for i, ms in enumerate(horizons):
        ref = cn.condense_tv(ms)
        for nm in g.EVO_NAMES:
            scale = max(1.0, float(np.abs(ref[nm]).max()))
            assert evo[nm][i].shape == ref[nm].shape
            assert np.abs(evo[nm][i] - ref[nm]).max() <= 1e-11 * scale, (i, nm)
    assert m.condense_device(N) > 0
    with pytest.raises(gpu.MldGpuError):
        m.condense(N + 1)                       # a time-varying handle has exactly N step models per horizon
    m.close()


def test_single_step_horizon_and_argument_checks():
    wl = syn.make_workload("cfg1", batch=1)
    ag = wl["agents"][0]
    m = gpu.GpuModel([[ag["mats"]]], ag["dims"], time_varying=True)
    evo, ref = m.condense(1), cn.condense(ag["mats"], 1)
    for nm in g.EVO_NAMES:
        assert np.abs(evo[nm][0] - ref[nm]).max() <= 1e-12 * max(1.0, np.abs(ref[nm]).max())
    m.close()
    with pytest.raises(ValueError):
        gpu.GpuModel([[ag["mats"]] * 2, [ag["mats"]] * 3], ag["dims"], time_varying=True)


@pytest.mark.parametrize("name,nb", [("cfg1", 4), ("cfg2", 4)])
def test_time_varying_solve_matches_oracle(name, nb):
    wl = syn.make_workload(name, batch=nb)
    ag = wl["agents"][0]
    d, N = ag["dims"], wl["N_tilde"]
    ms = _tv.step_models(ag["mats"], N, seed=21, strength=0.05)
    m = gpu.GpuModel([ms], d, time_varying=True)
    cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], N)
    p = gpu.GpuProblem(m, wl["N_p"], N, cost, max_nodes=20000)
    out = p.solve(ag["x0"], ag["omega"])
    tight = [tighten_np.tighten(mk, d, nu_l=d["nu_l"]) for mk in ms]
    sf_t = cn.standard_form(tight, ag["atoms"], wl["N_p"], N, nu_l=d["nu_l"])
    sf_o = cn.standard_form(ms, ag["atoms"], wl["N_p"], N, nu_l=d["nu_l"])
    n_opt = 0
    for s in range(nb):
        x0, om = ag["x0"][s], ag["omega"][s]
        h, q = cn.rhs(sf_t["evo"], x0, om), cn.lin_cost(sf_t["cost"], x0, om)
        r = cn.cost_const(sf_t["cost"]["const_terms"], x0, om)
        ref = orc.solve_milp(q, sf_t["G"], h, sf_t["lb"], sf_t["ub"], sf_t["is_bin"], max_nodes=20000, presolve=0)
        assert gpu._lib.STATUS_NAMES[int(out["status"][s])] == ref["status"], (s, out["status"][s], ref["status"])
        if ref["status"] != "optimal":
            continue
        n_opt += 1
        assert abs(out["obj"][s] - (ref["obj"] + r)) <= 1e-6 * max(1.0, abs(ref["obj"] + r)), (s, out["obj"][s], ref["obj"] + r)
        # certificate on the ORIGINAL (untightened) time-varying rows
        v = out["v"][s]
        G, ho = sf_o["G"], cn.rhs(sf_o["evo"], x0, om)
        bins = sf_o["is_bin"]
        assert np.all((v[bins] == 0) | (v[bins] == 1))
        assert np.all((G @ v - ho) / np.maximum(1.0, np.abs(G).max(axis=1)) <= 1e-6)
        qo = cn.lin_cost(sf_o["cost"], x0, om)
        ro = cn.cost_const(sf_o["cost"]["const_terms"], x0, om)
        assert abs(qo @ v + ro - out["obj"][s]) <= 1e-6 * max(1.0, abs(out["obj"][s]))
    assert n_opt >= 1
    p.close(); m.close()


def _thermo_steps(N):
    """a heater whose loss coefficient, gain and ambient offset follow a daily profile: one model per step"""
    import pyhybridcontrol_amd as phc
    out = []
    for k in range(N):
        a = 0.997 - 0.004 * k
        out.append(phc.MldModel(A=[[a]], B1=[[4.3 - 0.3 * k]], B4=[[-179.7]], b5=[[0.074 + 0.05 * k]],
                                E=[[1], [-1]], F1=[[0], [0]], Psi=[[-1, 0], [0, -1]], f5=[[65.0], [-50.0 - 1.0 * k]],
                                nu_l=1, ts=900))
    return out


def test_controller_time_varying_horizon_matches_highs_and_oracle():
    import pyhybridcontrol_amd as phc
    from scipy.optimize import milp, LinearConstraint, Bounds
    N = 5
    steps = _thermo_steps(N)
    ctrl = phc.MpcController(N_p=N - 1, mld_numeric_tilde=steps)
    assert ctrl.mld_numeric_k is steps[0] and len(ctrl.mld_numeric_tilde) == N
    price = np.array([1, 3, 3, 1, 1.0])
    atoms = dict(q_u=(price * 0.75).reshape(-1, 1), q_mu=[90.0, 90.0])
    ctrl.set_std_obj_atoms(**atoms)
    ctrl.build()
    x0, om = np.array([50.3]), np.array([.004, .012, 0, .009, .002])
    obj = ctrl.solve(0, x_k=x0, omega_tilde_k=om)
    ms = [m.as_mats() for m in steps]
    evo = cn.condense_tv(ms)
    assert np.allclose(ctrl.mld_evo_matrices.constraint["H_v_N_tilde"], evo["H_v"], rtol=0, atol=1e-12)
    assert np.allclose(ctrl.mld_evo_matrices.state_input["Phi_x_N_tilde"], evo["Phi_x"], rtol=0, atol=1e-12)
    sf = cn.standard_form(ms, {"q_u": atoms["q_u"], "q_mu": np.array(atoms["q_mu"]).reshape(-1, 1)}, N - 1, N, nu_l=1)
    h, q = cn.rhs(sf["evo"], x0, om), cn.lin_cost(sf["cost"], x0, om)
    r = cn.cost_const(sf["cost"]["const_terms"], x0, om)
    res = milp(q, constraints=LinearConstraint(sf["G"], -np.inf, h), integrality=sf["is_bin"].astype(int),
               bounds=Bounds(sf["lb"], sf["ub"]))
    assert res.status == 0 and abs(obj - (res.fun + r)) <= 1e-6 * max(1.0, abs(obj))
    ref = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"])
    assert ref["status"] == "optimal" and abs(obj - (ref["obj"] + r)) <= 1e-6 * max(1.0, abs(obj))
    v = ctrl.v_N_tilde.ravel()
    assert np.all(sf["G"] @ v <= h + 1e-7)
    # the time-invariant controller on step 0's model answers a different problem
    lti = phc.MpcController(steps[0], N_p=N - 1)
    lti.set_std_obj_atoms(**atoms); lti.build()
    assert abs(lti.solve(0, x_k=x0, omega_tilde_k=om) - obj) > 1e-3
    # receding horizon: the next step's models replace the horizon without touching cost atoms or state
    ctrl.sim_step_k(0)
    ctrl.mld_numeric_tilde = steps[1:] + steps[-1:]
    assert ctrl.build_required
    ctrl.build()
    obj2 = ctrl.solve(1, omega_tilde_k=om)
    evo2 = cn.condense_tv(ms[1:] + ms[-1:])
    assert np.allclose(ctrl.mld_evo_matrices.constraint["H_v_N_tilde"], evo2["H_v"], rtol=0, atol=1e-12)
    assert np.isfinite(obj2)
    with pytest.raises(ValueError):
        phc.MpcController(N_p=N - 1, mld_numeric_tilde=steps[:3])


def test_controller_time_varying_with_l1_atom_identical_steps_equals_time_invariant():
    """the augmented (epigraph) model of a time-varying horizon: with identical steps it must answer exactly as the
    time-invariant controller; with varying steps the returned point is checked against the original rows"""
    import pyhybridcontrol_amd as phc
    N = 5
    steps = _thermo_steps(N)
    price = np.array([1, 3, 3, 1, 1.0])
    x0, om = np.array([50.3]), np.array([.004, .012, 0, .009, .002])

    def run(**kw):
        c = phc.MpcController(N_p=N - 1, **kw)
        c.set_std_obj_atoms(q_u=(price * 0.75).reshape(-1, 1), q_mu=[90.0, 90.0], q_L1_x=0.01)
        c.build()
        return c, c.solve(0, x_k=x0, omega_tilde_k=om)

    c1, o1 = run(model=steps[0])
    c2, o2 = run(mld_numeric_tilde=[steps[0]] * N)
    assert abs(o1 - o2) <= 1e-9 * max(1.0, abs(o1)) and np.array_equal(c1.v_N_tilde, c2.v_N_tilde)
    c3, o3 = run(mld_numeric_tilde=steps)
    evo = cn.condense_tv([m.as_mats() for m in steps])
    v = c3.v_N_tilde.reshape(-1, 1)
    x = x0.reshape(-1, 1)
    rhs = evo["H_x"] @ x + evo["H_omega"] @ om.reshape(-1, 1) + evo["H_5"]
    assert np.all(evo["H_v"] @ v <= rhs + 1e-7)
    xs = evo["Phi_x"] @ x + evo["Gamma_v"] @ v + evo["Gamma_omega"] @ om.reshape(-1, 1) + evo["Gamma_5"]
    u, mu = v.reshape(N, 3)[:, 0], v.reshape(N, 3)[:, 1:]
    assert abs(o3 - (0.75 * price @ u + 90.0 * mu.sum() + 0.01 * np.abs(xs).sum())) <= 1e-6 * max(1.0, abs(o3))
