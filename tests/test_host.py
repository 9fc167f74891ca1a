"""Host-side mirror (no GPU): MldModel / MldInfo rules, objective-atom parsing and tiling against the
reference's golden vectors, controller error behaviour, C-ABI export surface."""
import os
import re

import numpy as np
import pytest

import _golden as g
import condense_np as cn
import pyhybridcontrol_amd as phc
from pyhybridcontrol_amd import _lib, host, objective_atoms as oa_mod, synthetic as syn


@pytest.mark.parametrize("path", g.case_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_mld_model_dims_and_types_match_reference(path):
    z, mats, dims, N_p, N_t = g.load_case(path)
    given = {k: v for k, v in mats.items() if v.size}
    if "C" not in given:
        given["C"] = mats["C"].reshape(dims["ny"], dims["nx"])
    m = phc.MldModel(given, nu_l=dims["nu_l"], nmu_l=dims["nmu_l"], ts=1)
    info = m.mld_info
    for k in ("nx", "nu", "ndelta", "nz", "nmu", "nomega", "ny", "n_constraints", "nu_l", "ndelta_l", "nz_l", "nmu_l", "nv_l"):
        assert info[k] == dims[k], (k, info[k], dims[k])
    assert info["nv"] == dims["nu"] + dims["ndelta"] + dims["nz"] + dims["nmu"]
    assert [str(t) for t in info["var_type_v"].ravel()] == [str(t) for t in z["var_type_v"]]
    for name in g.MATS:        # zero padding to (sys_dim, var_dim)
        if mats[name].size:
            assert np.array_equal(m[name], mats[name])


def test_mld_model_shape_errors_and_defaults():
    with pytest.raises(ValueError, match="must be a square matrix"):
        phc.MldModel(A=np.zeros((2, 3)))
    with pytest.raises(ValueError, match="Invalid matrix name"):
        phc.MldModel({"Q": [[1.0]]})
    with pytest.raises(ValueError, match="dimension"):
        phc.MldModel(A=np.eye(2), B1=np.zeros((3, 1)))
    with pytest.raises(ValueError, match="f5"):
        phc.MldModel(A=np.eye(1), E=[[1.0]])
    m = phc.MldModel(A=[[0.5]], B1=[[1.0]])
    assert np.array_equal(m.C, np.eye(1)) and m.mld_info.ny == 1           # C defaults to I
    assert m.D1.shape == (1, 1) and not m.D1.any() and m.F1.shape == (0, 1)
    with pytest.raises(ValueError):
        m.A[0, 0] = 1.0                                                     # read-only like the reference
    out = m.lsim_k(x_k=[2.0], u_k=[1.0])
    assert out["x_k1"][0, 0] == 2.0 and out["y"][0, 0] == 2.0


def test_objective_atoms_match_reference_weights():
    z = np.load(os.path.join(g.GDIR, "ref_objective_weights.npz"))
    dims = dict(nx=1, nu=1, ndelta=0, nz=0, nmu=2, nomega=1, ny=1, nc=2)
    tags = sorted({k.split("|")[0] for k in z.files if "|" in k and not k.startswith("spec")})
    for tag in tags:
        spec = {str(k): z["specval_%s|%s" % (tag, k)] for k in z["spec_" + tag]}
        atoms = phc.ObjectiveAtoms(dims, 4, 5, spec)
        ref_keys = [k for k in z.files if k.startswith(tag + "|")]
        assert len(atoms.weights) == len(ref_keys)
        for (var, atype, wtype, rate), w in atoms.weights.items():
            assert np.allclose(w, z["%s|%s|%s_%s%s" % (tag, var, atype, wtype, "_d" if rate else "")], rtol=0, atol=1e-14)


def test_objective_atom_parsing_rules_and_errors():
    assert oa_mod.parse_key("q_mu") == ("vector", "Linear", "mu", False, "")
    assert oa_mod.parse_key("Q_x_f") == ("matrix", "Quadratic", "x", False, "f")
    assert oa_mod.parse_key("q_Quadratic_y_N_p") == ("vector", "Quadratic", "y", False, "N_p")
    assert oa_mod.parse_key("q_L1_du") == ("vector", "L1", "u", True, "")
    assert oa_mod.parse_key("q_delta")[2:4] == ("delta", False)              # 'delta' is not a rate of 'elta'
    with pytest.raises(ValueError, match="is not valid"):
        oa_mod.parse_key("q_foo")
    dims = dict(nx=1, nu=1, ndelta=0, nz=0, nmu=2, nomega=1, ny=1, nc=2)
    l1 = phc.ObjectiveAtoms(dims, 2, 3, {"q_L1_u": -2.0, "q_mu": [1.0, 1.0]})
    assert not l1.to_cost()["lin_v"][[0, 3, 6]].any()                       # the L1 atom is not a linear weight on u ...
    blocks = l1.epigraph_blocks()                                              # ... it is an epigraph block |w| t, t >= +-u
    assert len(blocks) == 1 and blocks[0]["var"] == "u" and np.array_equal(blocks[0]["cost"], 2.0 * np.ones((3, 1)))
    rate = phc.ObjectiveAtoms(dims, 2, 3, {"q_du": 1.0, "q_L1_du": 2.0})     # rate atoms: not in the plain cost, not plain epigraph blocks
    assert not rate.to_cost()["lin_v"].any() and rate.epigraph_blocks() == []
    from pyhybridcontrol_amd import epigraph
    assert epigraph.rate_vars(rate.weights) == ["u"]
    with pytest.raises(NotImplementedError):
        epigraph.rate_vars(phc.ObjectiveAtoms(dims, 2, 3, {"q_dmu": [1.0, 1.0]}).weights)
    q22 = phc.ObjectiveAtoms(dims, 2, 3, {"q_L22_x": 3.0}).to_cost()           # L22 is the quadratic atom
    assert np.array_equal(q22["quad_x"], 9.0 * np.eye(3))
    assert not phc.ObjectiveAtoms(dims, 2, 3, {"q_u": 0.0}).weights           # all-zero weights are dropped


def test_cost_tiling_matches_oracle_assembly():
    wl = syn.make_workload("cfg2", batch=1, quadratic=True)
    ag = wl["agents"][0]
    d = ag["dims"]
    c = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
    sf = cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    evo = sf["evo"]
    Ws = c["quad_x"] + c["quad_x"].T
    q0 = c["lin_v"] + evo["Gamma_v"].T @ c["lin_x"] + (evo["Gamma_v"].T @ Ws @ evo["Gamma_5"])[:, 0]
    assert np.allclose(q0, sf["cost"]["q0"], rtol=0, atol=1e-12)
    assert np.allclose(evo["Gamma_v"].T @ Ws @ evo["Gamma_v"], sf["cost"]["P"], rtol=0, atol=1e-12)


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mldgpu.h")).read()
    declared = set(re.findall(r"\b(mld_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"mld_err", "mld_status"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert "mldgpu" in _lib.version()


def test_no_cpu_fallback_without_a_device():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    m = phc.MldModel(A=[[0.9]], B1=[[1.0]], E=[[1.0]], F1=[[0.0]], f5=[[10.0]], nu_l=1)
    ctrl = phc.MpcController(m, N_p=2)
    with pytest.raises(phc.ControllerBuildRequiredError):
        ctrl.solve(0, x_k=[1.0])
    with pytest.raises(phc.MldGpuError, match="no HIP device"):
        ctrl.build()


def test_fold_known_preserves_the_constraint_statement():
    """aux_resolve.fold_known: with u (and any known auxiliaries) moved into the disturbance channel, the folded
    model's rows evaluate to the original model's rows (models/mld_model.py:735-744) for arbitrary values"""
    from pyhybridcontrol_amd.aux_resolve import fold_known
    from pyhybridcontrol_amd import synthetic as syn
    wl = syn.make_workload("cfg2", batch=2)
    ag = wl["agents"][0]
    d, M = ag["dims"], ag["mats"]
    rng = np.random.default_rng(3)
    g = lambda mats, k, r, c: np.zeros((r, c)) if mats.get(k) is None or np.size(mats[k]) == 0 else np.asarray(mats[k], float).reshape(r, c)

    def rows(mats, dd, x, u, dl, z, mu, om):
        y = g(mats, "C", dd["ny"], dd["nx"]) @ x + g(mats, "D1", dd["ny"], dd["nu"]) @ u + g(mats, "D2", dd["ny"], dd["ndelta"]) @ dl + \
            g(mats, "D3", dd["ny"], dd["nz"]) @ z + g(mats, "D4", dd["ny"], dd["nomega"]) @ om + g(mats, "d5", dd["ny"], 1)[:, 0]
        con = (g(mats, "E", dd["nc"], dd["nx"]) @ x + g(mats, "F1", dd["nc"], dd["nu"]) @ u + g(mats, "F2", dd["nc"], dd["ndelta"]) @ dl +
               g(mats, "F3", dd["nc"], dd["nz"]) @ z + g(mats, "F4", dd["nc"], dd["nomega"]) @ om + g(mats, "G", dd["nc"], dd["ny"]) @ y +
               g(mats, "Psi", dd["nc"], dd["nmu"]) @ mu - g(mats, "f5", dd["nc"], 1)[:, 0])
        x1 = g(mats, "A", dd["nx"], dd["nx"]) @ x + g(mats, "B1", dd["nx"], dd["nu"]) @ u + g(mats, "B2", dd["nx"], dd["ndelta"]) @ dl + \
            g(mats, "B3", dd["nx"], dd["nz"]) @ z + g(mats, "B4", dd["nx"], dd["nomega"]) @ om + g(mats, "b5", dd["nx"], 1)[:, 0]
        return con, x1
    x, u, dl, z, mu, om = (rng.normal(size=d[k]) for k in ("nx", "nu", "ndelta", "nz", "nmu", "nomega"))
    for unknown in (("delta", "z", "mu"), ("z", "mu"), ("mu",), ("delta",)):
        m2, d2, known = fold_known(M, d, unknown)
        assert d2["nu"] == 0 and known[0] == "u"
        vals = dict(u=u, delta=dl, z=z, mu=mu)
        om2 = np.concatenate([om] + [vals[k] for k in known])
        assert om2.size == d2["nomega"]
        e = np.zeros(0)
        c2, x2 = rows(m2, d2, x, e, dl if "delta" in unknown else e, z if "z" in unknown else e, mu if "mu" in unknown else e, om2)
        c1, x1 = rows(M, d, x, u, dl, z, mu, om)
        assert np.allclose(c1, c2, rtol=0, atol=1e-9 * max(1.0, np.abs(c1).max()))
        assert np.allclose(x1, x2, rtol=0, atol=1e-9 * max(1.0, np.abs(x1).max()))


def test_lsim_k_argument_rules_and_sim_log_frame():
    """ParNotSet / None rules of lsim_k (models/mld_model.py:647-683) that need no solve, and the DataFrame layout
    of MldSimLog.get_concat_log (controllers/controller_base.py:116-146)"""
    m = phc.MldModel(A=[[0.5]], B1=[[1.0]], B4=[[2.0]])
    with pytest.raises(ValueError, match="omega_k"):
        m.lsim_k(x_k=[1.0], u_k=[1.0])
    with pytest.raises(ValueError, match="u_k"):
        m.lsim_k(x_k=[1.0], omega_k=[0.0])
    with pytest.raises(ValueError, match="not both"):
        m.lsim_k(x_k=[1.0], v_k=[1.0], u_k=[1.0], omega_k=[0.0])
    out = m.lsim_k(x_k=[2.0], u_k=None, omega_k=[1.0])                 # None means zeros
    assert out["x_k1"][0, 0] == 3.0 and out["u"][0, 0] == 0.0
    assert m.lsim_k(x_k=[2.0], v_k=[1.0], omega_k=[0.0])["x_k1"][0, 0] == 2.0
    assert repr(phc.ParNotSet) == "ParNotSet" and not phc.ParNotSet
    from pyhybridcontrol_amd.controllers import MldSimLog
    log = MldSimLog()
    log.set_sim_k(0, dict(x=[1.0, 2.0], u=[1.0]))
    log[2] = dict(x=[3.0, 4.0])
    log.update_sim_k(2, cost=5.0)
    with pytest.raises(ValueError, match="shape"):
        log.update_sim_k(2, x=[1.0])
    with pytest.raises(TypeError):
        log.update_sim_k(3, sim_k=[1, 2])
    df = log.get_concat_log()
    assert list(df.index) == [0, 2] and df.index.name == "k" and list(df.columns.names) == ["var_names", "var_index"]
    assert df[("x", 1)].tolist() == [2.0, 4.0] and np.isnan(df[("u", 0)][2]) and np.isnan(df[("cost", 0)][0])
    assert log.get_concat_log("mpc").columns[0] == ("mpc", "x", 0)


def test_epigraph_augmentation_equals_explicit_norm_formulation():
    """epigraph.py: the augmented MLD model + lifted linear cost solves  min c'v + sum_k ||w_k o u_k||_1 + sum_k ||W0 x_k||_1
    -- checked with HiGHS against the problem written down directly from the ORIGINAL condensed matrices with its own
    epigraph variables (what cvxpy builds for the reference's L1 atoms, objective_atoms.py:334-345)."""
    from scipy.optimize import milp, LinearConstraint, Bounds
    from pyhybridcontrol_amd import epigraph
    wl = syn.make_workload("cfg2", batch=1)
    ag = wl["agents"][0]
    d, N_p, N = ag["dims"], 5, 6
    rng = np.random.default_rng(12)
    w_u = rng.uniform(0.5, 2.0, size=(N * d["nu"], 1)) * rng.choice([-1.0, 1.0], size=(N * d["nu"], 1))
    W0 = rng.normal(size=(d["nx"], d["nx"])) * 1e-3
    atoms = {"q_z": rng.uniform(0.01, 0.1, size=(N * d["nz"], 1)), "q_mu": np.full((d["nmu"], 1), 5.0),
             "q_L1_u": w_u * 0.01, "Q_Linf_x": W0}
    w_u = w_u * 0.01
    oa = phc.ObjectiveAtoms(d, N_p, N, atoms)
    blocks = oa.epigraph_blocks()
    assert [b["var"] for b in blocks] == ["u", "x"]
    mats2, d2, nt = epigraph.augment(ag["mats"], d, blocks)
    assert nt == d["nu"] + d["nx"] and d2["nc"] == d["nc"] + 2 * nt
    cost = oa.to_cost(); cost.pop("_omega_atoms")
    lifted, vmap = epigraph.lift_cost(cost, d, d2, N, blocks)
    x0, om = ag["x0"][0], ag["omega"][0][:N * d["nomega"]]
    # (1) the augmented problem through the numpy restatement
    ev2 = cn.condense(mats2, N)
    G2, nv2 = ev2["H_v"], d2["nu"] + d2["ndelta"] + d2["nz"] + d2["nmu"]
    h2 = (ev2["H_x"] @ x0 + ev2["H_omega"] @ om + ev2["H_5"][:, 0])
    q2 = lifted["lin_v"] + ev2["Gamma_v"].T @ np.zeros(N * d["nx"])
    lb2, ub2, bin2 = np.full(N * nv2, -np.inf), np.full(N * nv2, np.inf), np.zeros(N * nv2, bool)
    for k in range(N):
        o = k * nv2
        lb2[o:o + d["nu"] + d["ndelta"]] = 0; ub2[o:o + d["nu"] + d["ndelta"]] = 1; bin2[o:o + d["nu"] + d["ndelta"]] = True
        lb2[o + nv2 - d["nmu"]:o + nv2] = 0
    r2 = milp(q2, constraints=LinearConstraint(G2, -np.inf, h2), bounds=Bounds(lb2, ub2), integrality=bin2.astype(int))
    # (2) written down directly: original rows, explicit epigraph variables tu (N nu) and tx (N nx)
    ev = cn.condense(ag["mats"], N)
    nv, n = d["nu"] + d["ndelta"] + d["nz"] + d["nmu"], N * (d["nu"] + d["ndelta"] + d["nz"] + d["nmu"])
    Su = np.zeros((N * d["nu"], n))
    for k in range(N):
        Su[k * d["nu"]:(k + 1) * d["nu"], k * nv:k * nv + d["nu"]] = np.eye(d["nu"])
    Wx = np.kron(np.eye(N), W0)
    xaff = ev["Phi_x"] @ x0 + ev["Gamma_omega"] @ om + ev["Gamma_5"][:, 0]
    nu_t, nx_t = N * d["nu"], N * d["nx"]
    A = np.block([[ev["H_v"], np.zeros((ev["H_v"].shape[0], nu_t + nx_t))],
                  [Su, -np.eye(nu_t), np.zeros((nu_t, nx_t))], [-Su, -np.eye(nu_t), np.zeros((nu_t, nx_t))],
                  [Wx @ ev["Gamma_v"], np.zeros((nx_t, nu_t)), -np.eye(nx_t)], [-Wx @ ev["Gamma_v"], np.zeros((nx_t, nu_t)), -np.eye(nx_t)]])
    hh = ev["H_x"] @ x0 + ev["H_omega"] @ om + ev["H_5"][:, 0]
    b = np.concatenate([hh, np.zeros(2 * nu_t), -Wx @ xaff, Wx @ xaff])
    c = np.concatenate([np.asarray(cost["lin_v"]).ravel(), np.abs(w_u[:, 0]), np.ones(nx_t)])
    lb = np.concatenate([lb2[vmap], np.full(nu_t + nx_t, -np.inf)]); ub = np.concatenate([ub2[vmap], np.full(nu_t + nx_t, np.inf)])
    isb = np.concatenate([bin2[vmap], np.zeros(nu_t + nx_t, bool)])
    r1 = milp(c, constraints=LinearConstraint(A, -np.inf, b), bounds=Bounds(lb, ub), integrality=isb.astype(int))
    assert r1.status == 0 and r2.status == 0
    assert abs(r1.fun - r2.fun) <= 1e-7 * max(1.0, abs(r1.fun)), (r1.fun, r2.fun)
    # the norms evaluated at the augmented solution equal the auxiliaries' cost
    v = r2.x[vmap]
    xs = ev["Gamma_v"] @ v + xaff
    norm_cost = np.abs(w_u[:, 0] * (Su @ v)).sum() + np.abs(Wx @ xs).sum()
    assert abs(np.asarray(cost["lin_v"]).ravel() @ v + norm_cost - r2.fun) <= 1e-7 * max(1.0, abs(r2.fun))


def test_rate_augmentation_equals_explicit_difference_formulation():
    """epigraph.augment_rates: lag states + rate outputs turn  sum_k w_k |u_k - u_{k-1}| + sum_k q (z_k - z_{k-1})^2-type
    atoms into ordinary atoms; checked with HiGHS (MILP, L1 + linear rate atoms) against the problem written with
    explicit difference variables on the ORIGINAL condensed matrices, including the value before the horizon."""
    from scipy.optimize import milp, LinearConstraint, Bounds
    from pyhybridcontrol_amd import epigraph
    wl = syn.make_workload("cfg2", batch=1)
    ag = wl["agents"][0]
    d, N_p, N = ag["dims"], 5, 6
    rng = np.random.default_rng(21)
    nu = d["nu"]
    w_du = rng.uniform(0.05, 0.2, size=(N * nu, 1))
    q_dx = rng.normal(size=(d["nx"], 1)) * 1e-3
    atoms = {"q_z": rng.uniform(0.01, 0.1, size=(N * d["nz"], 1)), "q_mu": np.full((d["nmu"], 1), 5.0),
             "q_L1_du": w_du, "q_dx": q_dx}
    oa = phc.ObjectiveAtoms(d, N_p, N, atoms)
    rv = epigraph.rate_vars(oa.weights)
    assert rv == ["x", "u"]
    mats1, d1, info = epigraph.augment_rates(ag["mats"], d, rv)
    assert d1["nx"] == d["nx"] + d["nx"] + nu and d1["ny"] == d["ny"] + d["nx"] + nu
    cost = oa.to_cost(); cost.pop("_omega_atoms")
    c1 = epigraph.lift_xy_cost(cost, d, d1, N)
    blocks = epigraph.rate_cost_and_blocks(oa.weights, d1, info, N, c1)
    assert len(blocks) == 1 and blocks[0]["var"] == "y"
    mats2, d2, nt = epigraph.augment(mats1, d1, blocks)
    lifted, vmap = epigraph.lift_cost(c1, d1, d2, N, blocks)
    x0, om = ag["x0"][0], ag["omega"][0][:N * d["nomega"]]
    u_prev, x_prev = np.array([1.0, 0.0, 1.0]), x0 + 0.7
    x_ext = np.concatenate([x0, x_prev, u_prev])                   # lag states in rate_vars order: x then u
    # (1) augmented model through the numpy restatement
    ev2 = cn.condense(mats2, N)
    nv2 = d2["nu"] + d2["ndelta"] + d2["nz"] + d2["nmu"]
    h2 = ev2["H_x"] @ x_ext + ev2["H_omega"] @ om + ev2["H_5"][:, 0]
    yaff = ev2["L_x"] @ x_ext + ev2["L_omega"] @ om + ev2["L_5"][:, 0]
    q2 = lifted["lin_v"] + ev2["L_v"].T @ lifted["lin_y"]
    const2 = lifted["lin_y"] @ yaff
    lb2, ub2, bin2 = np.full(N * nv2, -np.inf), np.full(N * nv2, np.inf), np.zeros(N * nv2, bool)
    for k in range(N):
        o = k * nv2
        lb2[o:o + nu + d["ndelta"]] = 0; ub2[o:o + nu + d["ndelta"]] = 1; bin2[o:o + nu + d["ndelta"]] = True
        lb2[o + nv2 - d["nmu"]:o + nv2] = 0
    r2 = milp(q2, constraints=LinearConstraint(ev2["H_v"], -np.inf, h2), bounds=Bounds(lb2, ub2), integrality=bin2.astype(int))
    # (2) written down directly on the original matrices: t >= +-(u_k - u_{k-1}),  q'(x_k - x_{k-1}) telescopes per step
    ev = cn.condense(ag["mats"], N)
    nv, n = d["nu"] + d["ndelta"] + d["nz"] + d["nmu"], N * (d["nu"] + d["ndelta"] + d["nz"] + d["nmu"])
    Su = np.zeros((N * nu, n))
    for k in range(N):
        Su[k * nu:(k + 1) * nu, k * nv:k * nv + nu] = np.eye(nu)
    Dm = np.eye(N * nu) - np.kron(np.eye(N, k=-1), np.eye(nu))      # u_k - u_{k-1}
    off = np.zeros(N * nu); off[:nu] = -u_prev
    xaff = ev["Phi_x"] @ x0 + ev["Gamma_omega"] @ om + ev["Gamma_5"][:, 0]
    Dx = np.eye(N * d["nx"]) - np.kron(np.eye(N, k=-1), np.eye(d["nx"]))
    offx = np.zeros(N * d["nx"]); offx[:d["nx"]] = -x_prev
    qx = np.tile(q_dx[:, 0], N)
    A = np.block([[ev["H_v"], np.zeros((ev["H_v"].shape[0], N * nu))], [Dm @ Su, -np.eye(N * nu)], [-Dm @ Su, -np.eye(N * nu)]])
    b = np.concatenate([ev["H_x"] @ x0 + ev["H_omega"] @ om + ev["H_5"][:, 0], -off, off])
    c = np.concatenate([np.asarray(cost["lin_v"]).ravel() + (qx @ Dx) @ ev["Gamma_v"], w_du[:, 0]])
    const1 = qx @ (Dx @ xaff + offx)
    lb = np.concatenate([lb2[vmap], np.full(N * nu, -np.inf)]); ub = np.concatenate([ub2[vmap], np.full(N * nu, np.inf)])
    isb = np.concatenate([bin2[vmap], np.zeros(N * nu, bool)])
    r1 = milp(c, constraints=LinearConstraint(A, -np.inf, b), bounds=Bounds(lb, ub), integrality=isb.astype(int))
    assert r1.status == 0 and r2.status == 0
    assert abs((r1.fun + const1) - (r2.fun + const2)) <= 1e-7 * max(1.0, abs(r1.fun + const1)), (r1.fun + const1, r2.fun + const2)


def test_fused_grid_model_equals_sequential_evaluation():
    """compose.fuse: devices + grid as ONE MLD system (the numeric counterpart of the example's cvxpy-level composition,
    micro_grid_agents.py:625-709): state update, outputs and constraint residuals of the fused system equal evaluating
    the devices first and feeding their outputs into the grid's disturbance channel"""
    from pyhybridcontrol_amd import compose
    rng = np.random.default_rng(8)

    def rand_model(nx, nu, nd, nz, nmu, nw, ny, nc, nu_l):
        d = dict(nx=nx, nu=nu, ndelta=nd, nz=nz, nmu=nmu, nomega=nw, ny=ny, nc=nc, nu_l=nu_l, nmu_l=0)
        shp = dict(A=(nx, nx), B1=(nx, nu), B2=(nx, nd), B3=(nx, nz), B4=(nx, nw), b5=(nx, 1), C=(ny, nx), D1=(ny, nu), D2=(ny, nd),
                   D3=(ny, nz), D4=(ny, nw), d5=(ny, 1), E=(nc, nx), F1=(nc, nu), F2=(nc, nd), F3=(nc, nz), F4=(nc, nw), f5=(nc, 1),
                   G=(nc, ny), Psi=(nc, nmu))
        return {k: rng.normal(size=v) for k, v in shp.items()}, d

    devs = [rand_model(2, 1, 1, 1, 2, 1, 1, 4, 1), rand_model(1, 2, 0, 1, 1, 2, 1, 3, 2), rand_model(1, 1, 0, 0, 0, 1, 2, 2, 0)]
    grid = rand_model(1, 0, 1, 1, 1, 5, 1, 4, 0)            # 4 device outputs + 1 external disturbance
    mats, dims, lay = compose.fuse(devs, grid)
    assert dims["nx"] == 5 and dims["nu"] == 4 and dims["nu_l"] == 3 and dims["nomega"] == 1 + 2 + 1 + 1 and dims["ny"] == 5
    assert lay["u"][2] == (0, 1)                              # the continuous-input device comes first: binaries stay trailing

    def ev(m, d, x, u, dl, z, mu, w):
        y = m["C"] @ x + m["D1"] @ u + m["D2"] @ dl + m["D3"] @ z + m["D4"] @ w + m["d5"][:, 0]
        x1 = m["A"] @ x + m["B1"] @ u + m["B2"] @ dl + m["B3"] @ z + m["B4"] @ w + m["b5"][:, 0]
        r = m["E"] @ x + m["F1"] @ u + m["F2"] @ dl + m["F3"] @ z + m["F4"] @ w + m["G"] @ y + m["Psi"] @ mu - m["f5"][:, 0]
        return x1, y, r
    subs = devs + [grid]
    vals = [{v: rng.normal(size=d[k]) for v, k in (("x", "nx"), ("u", "nu"), ("delta", "ndelta"), ("z", "nz"), ("mu", "nmu"), ("omega", "nomega"))}
            for _, d in subs]
    outs = [ev(m, d, *(vals[i][v] for v in ("x", "u", "delta", "z", "mu", "omega"))) for i, (m, d) in enumerate(devs)]
    wg = np.concatenate([o[1] for o in outs] + [vals[3]["omega"][4:]])
    outs.append(ev(grid[0], grid[1], vals[3]["x"], vals[3]["u"], vals[3]["delta"], vals[3]["z"], vals[3]["mu"], wg))
    stack = {}
    for v, key in (("x", "nx"), ("u", "nu"), ("delta", "ndelta"), ("z", "nz"), ("mu", "nmu"), ("omega", "nomega")):
        a = np.zeros(dims[key])
        for i in range(4):
            o, k = lay[v][i]
            a[o:o + k] = vals[i][v][4:] if (v == "omega" and i == 3) else vals[i][v]
        stack[v] = a
    X1, Y, R = ev(mats, dims, stack["x"], stack["u"], stack["delta"], stack["z"], stack["mu"], stack["omega"])
    for i in range(4):
        for full, part, name in ((X1, outs[i][0], "x"), (Y, outs[i][1], "y"), (R, outs[i][2], "c")):
            o, k = lay[name][i]
            assert np.allclose(full[o:o + k], part, rtol=0, atol=1e-12 * max(1.0, np.abs(part).max() if part.size else 1.0)), (i, name)
    with pytest.raises(ValueError):
        compose.fuse(devs, rand_model(1, 0, 0, 0, 0, 2, 1, 1, 0))        # grid with too few disturbance inputs


def test_check_numeric_tilde_validates_the_step_models():
    """time-varying horizon (mld_numeric_tilde): one numeric model per step, all with step 0's shapes and types"""
    import pyhybridcontrol_amd as phc
    from pyhybridcontrol_amd.controllers import check_numeric_tilde
    mk = lambda a, **kw: phc.MldModel(A=[[a]], B1=[[1.0]], E=[[1.0]], F1=[[0.0]], f5=[[2.0]], **kw)
    steps = [mk(0.9), mk(0.8), mk(0.7)]
    out = check_numeric_tilde(steps, 3)
    assert [m["A"][0, 0] for m in out] == [0.9, 0.8, 0.7]
    with pytest.raises(ValueError, match="one model per horizon step"):
        check_numeric_tilde(steps, 4)
    with pytest.raises(ValueError, match="differs from step 0"):
        check_numeric_tilde([mk(0.9), mk(0.8, nu_l=1), mk(0.7)], 3)
    two_rows = phc.MldModel(A=[[0.9]], B1=[[1.0]], E=[[1.0], [-1.0]], F1=[[0.0], [0.0]], f5=[[2.0], [2.0]])
    with pytest.raises(ValueError, match="differs from step 0"):
        check_numeric_tilde([mk(0.9), two_rows, mk(0.7)], 3)


def test_schedule_params_make_the_step_models_of_a_horizon():
    """gen_schedule_params_tilde / get_mld_numeric_tilde (models/mld_model.py:1181-1227) over a numeric model factory"""
    import pyhybridcontrol_amd as phc
    ps = dict(a=0.9, b=1.0)
    make = lambda p: phc.MldModel(A=[[p["a"]]], B1=[[p["b"]]], E=[[1.0]], F1=[[0.0]], f5=[[2.0]])
    assert phc.gen_schedule_params_tilde(3, ps) is None
    sched = phc.gen_schedule_params_tilde(3, ps, dict(a=[0.9, 0.8, 0.7]), b=[1.0, 2.0, 3.0], ignored=[1, 2, 3])
    assert sched == [dict(a=0.9, b=1.0), dict(a=0.8, b=2.0), dict(a=0.7, b=3.0)]
    with pytest.raises(ValueError, match="needs to be present in param_struct"):
        phc.gen_schedule_params_tilde(3, ps, dict(c=[1, 2, 3]))
    with pytest.raises(ValueError, match="must be equal to N_tilde"):
        phc.gen_schedule_params_tilde(3, ps, dict(a=[1, 2]))
    tilde = phc.get_mld_numeric_tilde(make, 3, ps, sched)
    assert [(m["A"][0, 0], m["B1"][0, 0]) for m in tilde] == [(0.9, 1.0), (0.8, 2.0), (0.7, 3.0)]
    same = phc.get_mld_numeric_tilde(make, 3, ps)
    assert same[0] is same[1] is same[2] and same[0]["A"][0, 0] == 0.9
    with pytest.raises(ValueError, match="must be equal to N_tilde"):
        phc.get_mld_numeric_tilde(make, 4, ps, sched)


def test_matmul_scalar_broadcast_rule():
    """utils/matrix_utils.py:20-28: `matmul` multiplies elementwise when either operand is scalar-like (all dims 1), else `@`"""
    from pyhybridcontrol_amd.objective_atoms import matmul, is_scalar_like, atleast_2d_col
    A = np.arange(6.0).reshape(3, 2)
    assert is_scalar_like(np.ones((1, 1))) and is_scalar_like(2.5) and is_scalar_like(np.ones((1, 1, 1))) and not is_scalar_like(np.ones((1, 2)))
    assert np.array_equal(matmul(A, np.array([[2.0]])), 2.0 * A)            # (3,2) "times" a 1x1 matrix broadcasts; A @ [[2.]] would raise
    assert np.array_equal(matmul(np.array([[2.0]]), A), 2.0 * A)
    assert np.array_equal(matmul(A, atleast_2d_col([1.0, -1.0])), A @ np.array([[1.0], [-1.0]]))
    assert np.array_equal(matmul(3.0, A), 3.0 * A)
    with pytest.raises(ValueError):
        matmul(A, np.ones((3, 1)))                                           # a real shape mismatch still fails


def test_open_nodes_of_a_stopped_depth_first_search_partition_what_is_left():
    """sub-tree hand-off: a depth-first search over all assignments of 7 binaries with the kernel's stack protocol (variable,
    first value, "sibling accounted for"), stopped after every possible number of leaves; the nodes gpu.expand_open_nodes reads off the stack
    must be pairwise disjoint and cover exactly the leaves not visited yet -- also under fixings the search itself ran with"""
    import itertools
    from pyhybridcontrol_amd.gpu import expand_open_nodes
    rng = np.random.default_rng(11)
    nb = 7
    pos = np.arange(nb)                                      # decision-vector index == binary position in this toy
    for trial in range(6):
        fix0 = np.full(nb, 255, np.uint8)
        for j in rng.choice(nb, size=trial % 3, replace=False):
            fix0[j] = rng.integers(0, 2)
        free = [j for j in range(nb) if fix0[j] == 255]
        total = 2 ** len(free)
        for stop_after in range(0, total + 1, max(1, total // 13)):
            visited, pruned, stack = [], [], []                 # stack entries: [var, current value, sibling accounted for]
            state = dict(stopped=None)

            def assignment():
                a = fix0.copy()
                for v_, val_, _ in stack:
                    a[v_] = val_
                return a

            def dfs():
                if state["stopped"] is not None:
                    return
                if len(visited) + len(pruned) >= stop_after:   # the node limit hits HERE: the current path is still open
                    state["stopped"] = [list(e) for e in stack]
                    return
                a = assignment()
                rest = [j for j in free if a[j] == 255]
                if not rest:
                    visited.append(tuple(int(x) for x in a))
                    return
                j = rest[int(rng.integers(len(rest)))]       # the branching variable differs from path to path, as in the solver
                first = int(rng.integers(0, 2))
                closed_other = bool(rng.random() < 0.2)      # penalty branching closes the sibling without a node of its own now and then
                if closed_other:                             # the closed sibling is pruned space from the moment the level is pushed
                    b = a.copy(); b[j] = 1 - first
                    for bits in itertools.product((0, 1), repeat=int((b == 255).sum())):
                        c = b.copy(); c[c == 255] = bits; pruned.append(tuple(int(x) for x in c))
                stack.append([j, first, 1 if closed_other else 0])
                dfs()
                if state["stopped"] is not None:
                    return
                if not closed_other:
                    stack[-1][1], stack[-1][2] = 1 - first, 1
                    dfs()
                    if state["stopped"] is not None:
                        return
                stack.pop()

            dfs()
            visited = visited + pruned
            if state["stopped"] is None:
                assert len(set(visited)) == total
                continue
            st = state["stopped"]
            nodes = expand_open_nodes(fix0, len(st), [e[0] for e in st], [e[1] for e in st], [e[2] for e in st], pos)
            covered = []
            for f in nodes:
                assert np.all((f == fix0) | (fix0 == 255)), "a node keeps the fixings the search ran under"
                for bits in itertools.product((0, 1), repeat=int((f == 255).sum())):
                    c = f.copy(); c[c == 255] = bits; covered.append(tuple(int(x) for x in c))
            assert len(covered) == len(set(covered)), "open nodes overlap"
            assert not (set(covered) & set(visited)), "an open node contains visited leaves"
            assert len(set(covered)) + len(set(visited)) == total, (trial, stop_after, len(set(covered)), len(set(visited)), total)
