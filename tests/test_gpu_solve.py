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


def test_cfg3_multi_model_batch_matches_oracle_or_reports_limit():
    """three distinct agents x 4 scenarios in one launch (model_idx), cfg3 shape, full branch-and-bound"""
    wl = syn.make_workload("cfg3", batch=4, n_agents=3)
    d = wl["agents"][0]["dims"]
    m = gpu.GpuModel([a["mats"] for a in wl["agents"]], d)
    cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, wl["N_p"], wl["N_tilde"]) for a in wl["agents"]])
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, max_nodes=3000)
    x0 = np.concatenate([a["x0"] for a in wl["agents"]])
    om = np.concatenate([a["omega"] for a in wl["agents"]])
    midx = np.repeat(np.arange(3), 4).astype(np.int32)
    out = p.solve(x0, om, midx)
    n_opt = 0
    for i in range(12):
        a, s = wl["agents"][i // 4], i % 4
        st = gpu._lib.STATUS_NAMES[int(out["status"][i])]
        assert st in ("optimal", "node_limit")
        if np.isfinite(out["obj"][i]):
            check_solution(a, wl, s, out["v"][i], out["obj"][i])
            assert out["lower_bound"][i] <= out["obj"][i] + 1e-6
        if st == "optimal":
            sf, q, h, r = _oracle_instance(a, wl, s)
            ref = orc.solve_milp(q, sf["G"], h, sf["lb"], sf["ub"], sf["is_bin"], max_nodes=50000, presolve=0)
            if ref["status"] == "optimal":
                assert abs(out["obj"][i] - (ref["obj"] + r)) <= 1e-6 * max(1.0, abs(ref["obj"] + r)), i
                n_opt += 1
    assert n_opt >= 11, n_opt          # (round 1 asked for 6 of 12)
    p.close(); m.close()


def test_relaxation_only_mode_with_fixed_binaries():
    """BASELINE cfg2: binaries fixed -> one LP per instance (no branching); equals the oracle's LP"""
    wl = syn.make_workload("cfg2", batch=8)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]))
    rng = np.random.Generator(np.random.PCG64(5))
    nb = p.n_bin
    fixed = np.zeros((8, nb), dtype=np.uint8)
    bins = np.where(p.is_bin)[0]
    isdelta = (bins % m.nv) == d["nu"]
    for s in range(8):
        om = ag["omega"][s].reshape(wl["N_tilde"], -1)
        u = (rng.random((wl["N_tilde"], d["nu"])) < 0.2).astype(np.uint8)
        y = u @ ag["params"]["P_h_Nom"] + om[:, -1]
        dl = (y >= 0).astype(np.uint8)                       # delta consistent with the sign of y
        fixed[s, ~isdelta] = u.ravel()
        fixed[s, isdelta] = dl
    out = p.solve(ag["x0"], ag["omega"], fixed_bin=fixed)
    for s in range(8):
        sf, q, h, r = _oracle_instance(ag, wl, s)
        lb, ub = sf["lb"].copy(), sf["ub"].copy()
        lb[bins] = ub[bins] = fixed[s]
        ref = orc.solve_milp(q, sf["G"], h, lb, ub, np.zeros_like(sf["is_bin"]), presolve=0, max_cuts=0, cut_rounds=0)
        assert gpu._lib.STATUS_NAMES[int(out["status"][s])] == ref["status"]
        if ref["status"] == "optimal":
            assert out["nodes"][s] == 1
            assert abs(out["obj"][s] - (ref["obj"] + r)) <= 1e-6 * max(1.0, abs(ref["obj"]))
            assert np.array_equal(out["v"][s][bins], fixed[s])
            check_solution(ag, wl, s, out["v"][s], out["obj"][s])
    p.close(); m.close()


def test_rhs_kernel_matches_oracle_including_scenario_row_min():
    wl = syn.make_workload("cfg2", batch=5, n_agents=2)
    d = wl["agents"][0]["dims"]
    m = gpu.GpuModel([a["mats"] for a in wl["agents"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], None)
    x0 = np.concatenate([a["x0"] for a in wl["agents"]])
    om = np.concatenate([a["omega"] for a in wl["agents"]])
    midx = np.repeat(np.arange(2), 5).astype(np.int32)
    h = p.rhs(x0, om, midx)
    for i in range(10):
        a = wl["agents"][i // 5]
        evo = cn.condense(a["mats"], wl["N_tilde"])
        ref = cn.rhs(evo, x0[i], om[i])
        assert np.abs(h[i] - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    # scenario form: row-min over scenario columns of H_omega @ Omega (controller_base.py:442-444)
    a = wl["agents"][0]
    evo = cn.condense(a["mats"], wl["N_tilde"])
    Om = a["omega"][:4]                                    # 4 scenarios, as columns
    hs = p.rhs(a["x0"][:1], Om[np.newaxis], np.zeros(1, np.int32), scenarios=4)
    ref = cn.rhs_scenarios(evo, a["x0"][0], Om.T)
    assert np.abs(hs[0] - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    p.close(); m.close()


def test_cost_assembly_kernel_matches_oracle_with_quadratic_atoms():
    wl = syn.make_workload("cfg2", batch=1, quadratic=True)
    ag = wl["agents"][0]
    d = ag["dims"]
    atoms = dict(ag["atoms"])
    atoms["q_Quadratic_y"] = 1e-4
    atoms["Q_u"] = 0.5 * np.eye(d["nu"])
    atoms["q_x"] = np.linspace(0.1, 0.3, d["nx"])
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(atoms, d, wl["N_p"], wl["N_tilde"]))
    got = p.cost_assemble()
    sf = cn.standard_form(ag["mats"], atoms, wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    for k in ("P", "q0", "Qx", "Qw"):
        ref = sf["cost"][k]
        assert np.abs(got[k][0] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), k
    p.close(); m.close()


def _miqp_oracle(agent, atoms, wl, s, fixed=None):
    d = agent["dims"]
    tm = tighten_np.tighten(agent["mats"], d, nu_l=d["nu_l"])
    sf = cn.standard_form(tm, atoms, wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    h = cn.rhs(sf["evo"], agent["x0"][s], agent["omega"][s])
    q = cn.lin_cost(sf["cost"], agent["x0"][s], agent["omega"][s])
    r = cn.cost_const(sf["cost"]["const_terms"], agent["x0"][s], agent["omega"][s])
    lb, ub, isb = sf["lb"].copy(), sf["ub"].copy(), sf["is_bin"].copy()
    if fixed is not None:
        bins = np.where(isb)[0]
        lb[bins] = ub[bins] = fixed
        isb = np.zeros_like(isb)
    ref = orc.solve_miqp(sf["cost"]["P"], q, sf["G"], h, lb, ub, isb, max_nodes=20000, presolve=0)
    return sf, q, h, r, ref


@pytest.mark.parametrize("name,nb", [("cfg1", 4), ("cfg2", 5)])
def test_miqp_solve_matches_oracle(name, nb):
    """quadratic atoms (the MIQP variant Q_x = 1e-3 I plus a quadratic weight on y / u): convex-QP relaxation at
    every node by simplicial decomposition, GPU vs the oracle's restatement of the same algorithm"""
    wl = syn.make_workload(name, batch=nb, quadratic=True)
    ag = wl["agents"][0]
    d = ag["dims"]
    atoms = dict(ag["atoms"])
    atoms["Q_u"] = 0.05 * np.eye(d["nu"])
    if d["ny"]:
        atoms["q_Quadratic_y"] = 1e-4 if name == "cfg2" else 1e-2
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(atoms, d, wl["N_p"], wl["N_tilde"]), max_nodes=20000)
    out = p.solve(ag["x0"], ag["omega"])
    for s in range(nb):
        sf, q, h, r, ref = _miqp_oracle(ag, atoms, wl, s)
        assert gpu._lib.STATUS_NAMES[int(out["status"][s])] == ref["status"] == "optimal", (s, out["status"][s], ref["status"])
        tot = ref["obj"] + r
        assert abs(out["obj"][s] - tot) <= 1e-6 * max(1.0, abs(tot)), (s, out["obj"][s], tot)
        v = out["v"][s]
        bins = sf["is_bin"]
        assert np.all((v[bins] == 0) | (v[bins] == 1))
        sf0 = cn.standard_form(ag["mats"], atoms, wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
        h0 = cn.rhs(sf0["evo"], ag["x0"][s], ag["omega"][s])
        rown = np.maximum(1.0, np.abs(sf0["G"]).max(axis=1))
        assert np.all((sf0["G"] @ v - h0) / rown <= 1e-6)
        assert abs(0.5 * v @ sf["cost"]["P"] @ v + q @ v + r - out["obj"][s]) <= 1e-6 * max(1.0, abs(tot))
    p.close(); m.close()


def test_qp_relaxation_only_mode_binaries_fixed():
    """BASELINE cfg2 as specified: QP relaxation kernel only, binaries fixed"""
    wl = syn.make_workload("cfg2", batch=6, quadratic=True)
    ag = wl["agents"][0]
    d = ag["dims"]
    atoms = dict(ag["atoms"])
    atoms["q_Quadratic_z"] = 1e-3
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(atoms, d, wl["N_p"], wl["N_tilde"]))
    rng = np.random.Generator(np.random.PCG64(9))
    bins = np.where(p.is_bin)[0]
    isdelta = (bins % m.nv) == d["nu"]
    fixed = np.zeros((6, p.n_bin), dtype=np.uint8)
    for s in range(6):
        om = ag["omega"][s].reshape(wl["N_tilde"], -1)
        u = (rng.random((wl["N_tilde"], d["nu"])) < 0.25).astype(np.uint8)
        y = u @ ag["params"]["P_h_Nom"] + om[:, -1]
        fixed[s, ~isdelta] = u.ravel()
        fixed[s, isdelta] = (y >= 0).astype(np.uint8)
    out = p.solve(ag["x0"], ag["omega"], fixed_bin=fixed)
    for s in range(6):
        sf, q, h, r, ref = _miqp_oracle(ag, atoms, wl, s, fixed=fixed[s])
        assert gpu._lib.STATUS_NAMES[int(out["status"][s])] == ref["status"] == "optimal"
        tot = ref["obj"] + r
        assert out["nodes"][s] == 1
        assert abs(out["obj"][s] - tot) <= 1e-6 * max(1.0, abs(tot)), (s, out["obj"][s], tot)
    p.close(); m.close()


def test_mpc_controller_end_to_end_known_answer():
    """the reference's call sequence: set_std_obj_atoms -> build -> solve/feedback -> sim_step_k (SURVEY 8c KAT)"""
    import pyhybridcontrol_amd as phc
    model = phc.MldModel(A=[[0.9970371127900564]], B1=[[4.298192277481107]], B4=[[-179.73320827515]],
                         b5=[[0.07407218024859108]], E=[[1], [-1]], F1=[[0], [0]], Psi=[[-1, 0], [0, -1]],
                         f5=[[65.0], [-50.0]], nu_l=1, ts=900)
    ctrl = phc.MpcController(model, N_p=4)
    assert ctrl.N_tilde == 5
    with pytest.raises(phc.ControllerBuildRequiredError):
        ctrl.solve(0, x_k=[50.3])
    price = np.array([1, 3, 3, 1, 1.0])
    ctrl.set_std_obj_atoms(q_u=(price * 0.75).reshape(-1, 1), q_mu=[90.0, 90.0])
    ctrl.build()
    obj = ctrl.solve(0, x_k=[50.3], omega_tilde_k=[.004, .012, 0, .009, .002])
    assert abs(obj - 1.5) < 1e-9
    assert np.array_equal(ctrl.v_N_tilde.reshape(5, 3)[:, 0], [1, 0, 0, 1, 0])
    vk = ctrl.feedback(0)
    assert vk["u"].shape == (1, 1) and vk["u"][0, 0] == 1.0 and vk["mu"].shape == (2, 1)
    x, y = ctrl.predicted_trajectory()
    evo = cn.condense(model.as_mats(), 5)
    assert np.allclose(ctrl.mld_evo_matrices.constraint["H_v_N_tilde"], evo["H_v"], rtol=0, atol=1e-12)
    assert ctrl.mld_evo_matrices.state_input["Gamma_v_N_p"].shape == (4, 15)
    cons = ctrl.gen_evo_constraints()
    assert np.all(cons.H_v @ ctrl.v_N_tilde <= cons.rhs + 1e-9)
    assert np.all(x[:, 0] >= 50.0 - 1e-6)
    sim = ctrl.sim_step_k(0)
    assert abs(sim["x_k1"][0, 0] - x[1, 0]) < 1e-9 and 0 in ctrl.sim_log
    assert ctrl.x_k[0, 0] == sim["x_k1"][0, 0]
    assert ctrl._solve_time_overall > 0 and ctrl._solve_time_solver > 0
    assert ctrl.solve(1, external_solve=7.0) == 7.0            # external_solve bypass (controller_base.py:536-538)
    ctrl.set_std_obj_atoms(q_u=price.reshape(-1, 1), q_mu=[90.0, 90.0])
    with pytest.raises(phc.ControllerBuildRequiredError):
        ctrl.solve(1)
    ctrl.build(sense="maximize")
    with pytest.raises(phc.ControllerSolverError, match="unbounded"):      # max 90 mu has no finite value
        ctrl.solve(1, MIPGap=1e-2)
    ctrl.set_std_obj_atoms(q_u=price.reshape(-1, 1))
    ctrl.build(sense="maximize")
    assert abs(ctrl.solve(1, MIPGap=1e-2) - price.sum()) < 1e-9 and np.all(ctrl.v_N_tilde.reshape(5, 3)[:, 0] == 1)
    # an infeasible instance -> ControllerSolverError (hard bound instead of the soft one)
    hard = phc.MldModel(A=[[1.0]], B1=[[0.0]], E=[[1.0], [-1.0]], F1=[[0.0], [0.0]], f5=[[1.0], [-2.0]], nu_l=1)
    c2 = phc.MpcController(hard, N_p=1)
    c2.set_std_obj_atoms(q_u=1.0)
    c2.build()
    with pytest.raises(phc.ControllerSolverError):
        c2.solve(0, x_k=[0.0])


def test_full_size_batch_properties_cfg3():
    """BASELINE cfg3 at full batch (1024): size-independent properties instead of per-instance oracle solves:
    every returned point is integer feasible for the original rows with the reported cost, lower bound <=
    objective, a second solve is bit-identical (idempotence), and solving the two halves separately returns
    the same per-instance results (sharding must not change any instance's arithmetic)."""
    wl = syn.make_workload("cfg3", batch=1024)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]),
                       max_nodes=300)
    out = p.solve(ag["x0"], ag["omega"])
    fin = np.isfinite(out["obj"])
    assert fin.all() and (out["status"] == 0).mean() >= 0.93, (fin.mean(), (out["status"] == 0).mean())     # exact gap, 300 nodes (measured 0.954; round 1 asked for > 0.5)
    sf = cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    G, bins = sf["G"], sf["is_bin"]
    rown = np.maximum(1.0, np.abs(G).max(axis=1))
    H = np.stack([cn.rhs(sf["evo"], ag["x0"][s], ag["omega"][s]) for s in range(1024)])
    V = out["v"]
    assert np.all(((V[fin][:, bins] == 0) | (V[fin][:, bins] == 1)))
    assert np.all((V[fin] @ G.T - H[fin]) / rown <= 1e-6)
    assert np.all(np.abs(V[fin] @ sf["cost"]["q0"] - out["obj"][fin]) <= 1e-6 * np.maximum(1, np.abs(out["obj"][fin])))
    assert np.all(out["lower_bound"][fin] <= out["obj"][fin] + 1e-6)
    out2 = p.solve(ag["x0"], ag["omega"])
    assert np.array_equal(out2["v"], V) and np.array_equal(out2["obj"], out["obj"]) and np.array_equal(out2["status"], out["status"])
    a = p.solve(ag["x0"][:512], ag["omega"][:512])
    b = p.solve(ag["x0"][512:], ag["omega"][512:])
    assert np.array_equal(np.concatenate([a["obj"], b["obj"]]), out["obj"])
    assert np.array_equal(np.concatenate([a["v"], b["v"]]), V)
    p.close(); m.close()


def test_maximum_size_cfg5_condense_and_solve_certificates():
    """BASELINE cfg5 shape (n_h=15, N_p=48: n=2303, 784 binaries, m=1764; 36 MB dictionary per slot, hot basis
    state only partly LDS-resident): condensing vs the numpy oracle and, for a few instances under a small node
    limit, the size-independent certificates (integer feasible for the original rows, reported cost, bound)."""
    wl = syn.make_workload("cfg5", batch=6)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    evo = m.condense(wl["N_tilde"])
    ref = cn.condense(ag["mats"], wl["N_tilde"])
    for k in ("Phi_x", "Gamma_v", "Gamma_omega", "Gamma_5", "H_x", "H_v", "H_omega", "H_5"):
        err = np.abs(evo[k][0] - ref[k]).max() / max(1.0, np.abs(ref[k]).max())
        assert err < 1e-11, (k, err)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]),
                       gap_rel=1e-2, max_nodes=40, max_pivots=6000)
    out = p.solve(ag["x0"], ag["omega"])
    assert set(np.unique(out["status"])) <= {0, 2}, out["status"]
    fin = np.isfinite(out["obj"])
    assert fin.sum() >= 4
    for s in np.where(fin)[0]:
        check_solution(ag, wl, s, out["v"][s], out["obj"][s])
        assert out["lower_bound"][s] <= out["obj"][s] + 1e-6
    p.close(); m.close()


def test_closed_loop_receding_horizon_matches_highs_every_step():
    """twelve MPC iterations in closed loop (solve -> feedback -> sim_step_k -> shifted disturbance forecast): at every step the
    GPU objective equals HiGHS on the same condensed problem, the applied input is the first step of the plan and the state the
    controller carries forward is the model's own evolution"""
    import pyhybridcontrol_amd as phc
    from scipy.optimize import milp, LinearConstraint, Bounds
    model = phc.MldModel(A=[[0.9970371127900564]], B1=[[4.298192277481107]], B4=[[-179.73320827515]],
                         b5=[[0.07407218024859108]], E=[[1], [-1]], F1=[[0], [0]], Psi=[[-1, 0], [0, -1]],
                         f5=[[65.0], [-50.0]], nu_l=1, ts=900)
    N_p, steps = 8, 12
    N = N_p + 1
    rng = np.random.default_rng(17)
    draw = rng.uniform(0.0, 0.02, steps + N)            # hot-water draw forecast
    price = 1.0 + 2.0 * (np.arange(steps + N) % 6 < 2)
    ctrl = phc.MpcController(model, N_p=N_p)
    sf0 = None
    x = np.array([51.0])
    for k in range(steps):
        q_u = (0.75 * price[k:k + N]).reshape(-1, 1)
        ctrl.set_std_obj_atoms(q_u=q_u, q_mu=[90.0, 90.0])
        ctrl.build()
        om = draw[k:k + N]
        obj = ctrl.solve(k, x_k=x, omega_tilde_k=om)
        sf = cn.standard_form(model.as_mats(), {"q_u": q_u, "q_mu": np.array([[90.0], [90.0]])}, N_p, N, nu_l=1)
        h, q = cn.rhs(sf["evo"], x, om), cn.lin_cost(sf["cost"], x, om)
        r = cn.cost_const(sf["cost"]["const_terms"], x, om)
        ref = milp(q, constraints=LinearConstraint(sf["G"], -np.inf, h), integrality=sf["is_bin"].astype(int),
                   bounds=Bounds(sf["lb"], sf["ub"]))
        assert ref.status == 0 and abs(obj - (ref.fun + r)) <= 1e-6 * max(1.0, abs(obj)), (k, obj, ref.fun + r)
        v = ctrl.v_N_tilde.ravel()
        assert np.all(sf["G"] @ v <= h + 1e-7)
        fb = ctrl.feedback(k)
        assert fb["u"][0, 0] == v[0] and fb["u"][0, 0] in (0.0, 1.0)
        sim = ctrl.sim_step_k(k)
        x_next = model["A"] @ x.reshape(1, 1) + model["B1"] * v[0] + model["B4"] * om[0] + model["b5"]
        assert abs(sim["x_k1"][0, 0] - x_next[0, 0]) <= 1e-9 and abs(ctrl.x_k[0, 0] - x_next[0, 0]) <= 1e-9
        x = np.array([x_next[0, 0]])
    assert sorted(ctrl.sim_log.keys()) == list(range(steps))
    assert 49.0 <= x[0] <= 66.0                                           # the thermostat band held (softly) over the run


def test_never_binding_rows_do_not_change_the_answer():
    """rows whose largest activity under the root bounds is below their right-hand side are not maintained by the pivots
    (s_mark_dead / oracle mark_dead): the same batch with the elimination switched off (solver option reserved bit 4) must
    give the same statuses and objectives, and a batch in which most rows are of that kind must still match HiGHS"""
    from scipy.optimize import milp, LinearConstraint, Bounds
    wl = syn.make_workload("cfg3", batch=96)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
    outs = []
    for flag in (0, 16):
        p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, max_nodes=2000, gap_rel=1e-4, reserved=flag)
        outs.append(p.solve(ag["x0"], ag["omega"]))
        p.close()
    a, b = outs
    assert np.array_equal(a["status"], b["status"])
    fin = (a["status"] == 0) & (b["status"] == 0)
    assert fin.sum() >= 88
    assert np.all(np.abs(a["obj"][fin] - b["obj"][fin]) <= 2e-4 * np.maximum(1.0, np.abs(b["obj"][fin])))     # both within the 1e-4 gap
    # how many rows the rule removes on this shape (numpy restatement of the rule on the tightened, un-scaled rows)
    sf = cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    G, lb, ub = sf["G"], sf["lb"], sf["ub"]
    pos, neg = np.maximum(G, 0), np.minimum(G, 0)
    with np.errstate(invalid="ignore"):
        act = np.where(pos > 0, pos * ub, 0).sum(1) + np.where(neg < 0, neg * lb, 0).sum(1)
    h0 = cn.rhs(sf["evo"], ag["x0"][0], ag["omega"][0]).ravel()
    assert (act <= h0 - 1e-7).mean() > 0.2
    # and the answers are HiGHS's (original rows)
    raw = cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    for s in range(4):
        h, q = cn.rhs(raw["evo"], ag["x0"][s], ag["omega"][s]), cn.lin_cost(raw["cost"], ag["x0"][s], ag["omega"][s])
        r = cn.cost_const(raw["cost"]["const_terms"], ag["x0"][s], ag["omega"][s])
        ref = milp(q, constraints=LinearConstraint(raw["G"], -np.inf, h), integrality=raw["is_bin"].astype(int),
                   bounds=Bounds(raw["lb"], raw["ub"]), options=dict(mip_rel_gap=1e-6))
        if a["status"][s] == 0 and ref.status == 0:
            assert abs(a["obj"][s] - (ref.fun + r)) <= 2e-4 * max(1.0, abs(ref.fun + r)), (s, a["obj"][s], ref.fun + r)
    m.close()
