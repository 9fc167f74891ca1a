"""Closed loop on device and per-call limits: mld_advance_batch (the reference's sim_step_k / lsim_k plant update,
controllers/controller_base.py:229-253, models/mld_model.py:647-699), mld_problem_set_opts (per-call solver kwargs,
controller_base.py:491-512) and the deep-cut tight-gap regression against HiGHS."""
import numpy as np
import pytest

import condense_np as cn
from pyhybridcontrol_amd import gpu, host, synthetic as syn

pytestmark = pytest.mark.gpu


def test_advance_batch_is_the_plant_update_and_forecast_shift():
    wl = syn.make_workload("cfg2", batch=6, n_agents=2)
    d = wl["agents"][0]["dims"]
    N = wl["N_tilde"]
    m = gpu.GpuModel([a["mats"] for a in wl["agents"]], d)
    cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, wl["N_p"], N) for a in wl["agents"]])
    p = gpu.GpuProblem(m, wl["N_p"], N, cost, gap_rel=1e-2, max_nodes=400)
    x0 = np.concatenate([a["x0"] for a in wl["agents"]])
    om = np.concatenate([a["omega"] for a in wl["agents"]])
    midx = np.repeat(np.arange(2), 6).astype(np.int32)
    out = p.solve(x0, om, midx)
    p.advance()
    x1, om1 = p.inputs()
    nv, nw = m.nv, d["nomega"]
    for b in range(12):
        mats = cn.pad_mats(wl["agents"][midx[b]]["mats"], cn.mld_dims(wl["agents"][midx[b]]["mats"]))
        v0 = out["v"][b][:nv]
        u, dl, z = v0[:d["nu"]], v0[d["nu"]:d["nu"] + d["ndelta"]], v0[d["nu"] + d["ndelta"]:d["nu"] + d["ndelta"] + d["nz"]]
        w0 = om[b].reshape(N, nw)[0]
        ref = mats["A"] @ x0[b] + mats["B1"] @ u + mats["B2"] @ dl + mats["B3"] @ z + mats["B4"] @ w0 + mats["b5"][:, 0]
        assert np.allclose(x1[b], ref, rtol=1e-13, atol=1e-12), b
        assert np.array_equal(om1[b].reshape(N, nw), np.roll(om[b].reshape(N, nw), -1, axis=0)), b
    # the next step solves the advanced data: identical to uploading it explicitly
    nxt = p.solve_resident(); got = p.download()
    ref = p.solve(x1, om1, midx)
    assert np.array_equal(got["status"], ref["status"]) and np.array_equal(got["obj"], ref["obj"])
    assert nxt["n_optimal"] + nxt["n_node_limit"] == 12
    p.close(); m.close()


def test_set_opts_changes_limits_without_rebuild_and_back():
    wl = syn.make_workload("cfg3", batch=16)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=400,
                       cut_rounds=0)      # (no cuts: with them every one of these instances is proven at its root and a node limit has nothing to limit)
    a = p.solve(ag["x0"], ag["omega"])
    p.set_opts(max_nodes=1)
    b = p.solve(ag["x0"], ag["omega"])
    assert b["nodes"].max() <= 1 + 3 * p.n_bin + 10 and (b["status"] == 2).sum() > (a["status"] == 2).sum()
    p.set_opts(MIPGap=1e-2, NodeLimit=400)
    c = p.solve(ag["x0"], ag["omega"])
    assert np.array_equal(a["obj"], c["obj"]) and np.array_equal(a["status"], c["status"])
    with pytest.raises(TypeError):
        p.set_opts(max_cuts=10)
    p.close(); m.close()


def test_solve_kwargs_are_per_call_and_keep_the_build_arguments():
    """solve(k, MIPGap=...) used to rebuild with default build() arguments (soft constraints back on) and to persist"""
    import pyhybridcontrol_amd as phc
    model = phc.MldModel(A=[[0.9970371127900564]], B1=[[4.298192277481107]], B4=[[-179.73320827515]],
                         b5=[[0.07407218024859108]], E=[[1], [-1]], F1=[[0], [0]], Psi=[[-1, 0], [0, -1]],
                         f5=[[65.0], [-50.0]], nu_l=1, ts=900)
    price = np.array([1, 3, 3, 1, 1.0])
    om = [.004, .012, 0, .009, .002]
    soft = phc.MpcController(model, N_p=4)
    soft.set_std_obj_atoms(q_u=(price * 0.75).reshape(-1, 1), q_mu=[1e-3, 1e-3])     # violating the soft bounds is nearly free
    soft.build()
    cheap = soft.solve(0, x_k=[50.3], omega_tilde_k=om)
    ctrl = phc.MpcController(model, N_p=4)
    ctrl.set_std_obj_atoms(q_u=(price * 0.75).reshape(-1, 1), q_mu=[1e-3, 1e-3])
    ctrl.build(disable_soft_constraints=True)
    hard = ctrl.solve(0, x_k=[50.3], omega_tilde_k=om)
    assert hard > cheap + 0.5, (hard, cheap)
    again = ctrl.solve(0, MIPGap=1e-2, NodeLimit=50)
    assert abs(hard - again) <= 1e-2 * abs(hard) + 1e-9, "solver kwargs must not re-enable the soft constraints"
    assert np.all(np.abs(ctrl.v_N_tilde.reshape(5, 3)[:, 1:]) <= 1e-8)
    assert ctrl._problem.opts.max_nodes == 50
    ctrl.solve(0)
    assert ctrl._problem.opts.max_nodes != 50, "per-call kwargs must not persist"


def test_deep_cut_loop_claims_match_highs():
    """round 1 withdrew deeper root cutting after ONE false optimality claim in this batch (23.1755 against 23.1641 at 12 rounds
    x 80 Gomory cuts, gap 1e-4).  With the tiny-pivot guard every proven objective must be within the gap of HiGHS's optimum."""
    from scipy.optimize import Bounds, LinearConstraint, milp
    nb = 96
    wl = syn.make_workload("cfg3", batch=nb)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
    raw = cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    for kw in (dict(cut_rounds=12, cuts_per_round=80, max_cuts=400), dict()):
        p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, max_nodes=2000, gap_rel=1e-4, **kw)
        out = p.solve(ag["x0"], ag["omega"])
        p.close()
        proven = np.where(out["status"] == 0)[0]
        assert proven.size >= 0.9 * nb, (kw, proven.size)
        for s in proven[:: 2 if kw else 4]:
            h, q = cn.rhs(raw["evo"], ag["x0"][s], ag["omega"][s]), cn.lin_cost(raw["cost"], ag["x0"][s], ag["omega"][s])
            r = cn.cost_const(raw["cost"]["const_terms"], ag["x0"][s], ag["omega"][s])
            ref = milp(q, constraints=LinearConstraint(raw["G"], -np.inf, h), integrality=raw["is_bin"].astype(int),
                       bounds=Bounds(raw["lb"], raw["ub"]), options=dict(mip_rel_gap=1e-7, time_limit=60))
            if ref.status != 0:
                continue
            rel = (out["obj"][s] - (ref.fun + r)) / max(1.0, abs(ref.fun + r))
            assert -1e-6 <= rel <= 2e-4, (kw, int(s), float(out["obj"][s]), float(ref.fun + r))
            assert out["lower_bound"][s] <= ref.fun + r + 1e-6 * max(1.0, abs(ref.fun + r))
    m.close()


def test_rhs_and_cost_on_the_matrix_cores_fp64_and_fp32():
    """K3 as one MFMA GEMM per model (batch = N dimension, instances of several models interleaved, ragged groups) against the
    numpy restatement and against the vector-ALU kernel (opts.reserved bit 7); MLD_F32 within the fp32 tolerance (1e-5 of the
    row's scale); K4's quadratic pull-back P = Gamma' (W + W') Gamma the same way."""
    from pyhybridcontrol_amd import _lib
    wl = syn.make_workload("cfg3", batch=150, n_agents=3, quadratic=True)
    d = wl["agents"][0]["dims"]
    N = wl["N_tilde"]
    m = gpu.GpuModel([a["mats"] for a in wl["agents"]], d)
    cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, wl["N_p"], N) for a in wl["agents"]])
    x0 = np.concatenate([a["x0"] for a in wl["agents"]])
    om = np.concatenate([a["omega"] for a in wl["agents"]])
    rng = np.random.Generator(np.random.PCG64(11))
    midx = np.repeat(np.arange(3), 150).astype(np.int32)
    order = rng.permutation(450)[:401]                     # interleaved models, ragged last groups
    x0, om, midx = x0[order], om[order], midx[order]
    evos = [cn.condense(a["mats"], N) for a in wl["agents"]]
    ref = np.stack([cn.rhs(evos[midx[b]], x0[b], om[b]) for b in range(401)])
    scale = np.maximum(1.0, np.abs(ref).max(axis=0))
    res = {}
    for name, kw in (("mfma64", dict()), ("valu", dict(reserved=128)), ("mfma32", dict(flags=_lib.MLD_F32))):
        p = gpu.GpuProblem(m, wl["N_p"], N, cost, **kw)
        res[name] = (p.rhs(x0, om, midx), p.cost_assemble())
        p.close()
    assert np.abs(res["mfma64"][0] - ref).max() / scale.max() <= 1e-12
    assert np.abs((res["mfma64"][0] - res["valu"][0]) / scale).max() <= 1e-12
    err32 = np.abs((res["mfma32"][0] - ref) / scale).max()
    assert 1e-12 < err32 <= 1e-5, err32                   # really fp32, and within the stated tolerance
    P64, Pv, P32 = res["mfma64"][1]["P"], res["valu"][1]["P"], res["mfma32"][1]["P"]
    ps = np.abs(Pv).max()
    assert ps > 0 and np.abs(P64 - Pv).max() <= 1e-12 * ps
    assert 1e-13 * ps < np.abs(P32 - Pv).max() <= 1e-5 * ps
    for k in ("q0", "Qx", "Qw"):
        s = max(1.0, np.abs(res["valu"][1][k]).max())
        assert np.abs(res["mfma64"][1][k] - res["valu"][1][k]).max() <= 1e-12 * s, k
        assert np.abs(res["mfma32"][1][k] - res["valu"][1][k]).max() <= 1e-5 * s, k
    m.close()


def test_gather_results_from_device_buffers_one_rank_communicator():
    """mld_gather_results over a one-rank RCCL communicator (all a one-GPU box can host): (objective, status, step-0 slice) rows
    equal the downloaded results"""
    from pyhybridcontrol_amd import _lib
    from pyhybridcontrol_amd.batch import RcclGather
    wl = syn.make_workload("cfg2", batch=9)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=300)
    out = p.solve(ag["x0"], ag["omega"])
    g = RcclGather(1, 0, RcclGather.unique_id())
    try:
        rows = g.gather_results(p)
    finally:
        _lib.load().mld_comm_destroy()
    assert rows.shape == (1, 9, 2 + m.nv)
    assert np.array_equal(rows[0, :, 0], out["obj"]) and np.array_equal(rows[0, :, 1], out["status"].astype(float))
    assert np.array_equal(rows[0, :, 2:], out["v"][:, :m.nv])
    p.close(); m.close()


def test_staged_input_sets_equal_direct_uploads():
    """mld_stage_inputs / mld_select_inputs: a staged scenario set solved in place equals the same set uploaded from the host"""
    wl = syn.make_workload("cfg2", batch=10, n_agents=2)
    d = wl["agents"][0]["dims"]
    N = wl["N_tilde"]
    m = gpu.GpuModel([a["mats"] for a in wl["agents"]], d)
    cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, wl["N_p"], N) for a in wl["agents"]])
    p = gpu.GpuProblem(m, wl["N_p"], N, cost, gap_rel=1e-2, max_nodes=400)
    midx = np.tile(np.arange(2), 10).astype(np.int32)
    sets = []
    for t in range(3):
        rng = np.random.Generator(np.random.PCG64(100 + t))
        sets.append(syn.make_scenarios(d["nx"], N, 20, rng))
    p.upload(sets[0][0], sets[0][1], midx)
    assert p.stage(np.stack([s[0] for s in sets]), np.stack([s[1] for s in sets])) == 3
    for t in (2, 0, 1):
        p.select(t)
        x, w = p.inputs()
        assert np.array_equal(x, sets[t][0]) and np.array_equal(w, sets[t][1])
        p.solve_resident(); got = p.download()
        ref = p.solve(sets[t][0], sets[t][1], midx)
        assert np.array_equal(got["obj"], ref["obj"]) and np.array_equal(got["status"], ref["status"]) and np.array_equal(got["v"], ref["v"])
    with pytest.raises(gpu.MldGpuError):
        p.select(3)
    p.close(); m.close()


def _fixed_pattern(ag, wl, p, d, rng, nb):
    """random heater schedules with the grid binary consistent with the sign of the tie flow (as test_gpu_solve does)"""
    bins = np.where(p.is_bin)[0]
    nv = d["nu"] + d["ndelta"] + d["nz"] + d["nmu"]
    isdelta = (bins % nv) == d["nu"]
    fixed = np.zeros((nb, p.n_bin), dtype=np.uint8)
    for s in range(nb):
        om = ag["omega"][s].reshape(wl["N_tilde"], -1)
        u = (rng.random((wl["N_tilde"], d["nu"])) < 0.2).astype(np.uint8)
        y = u @ ag["params"]["P_h_Nom"] + om[:, -1]
        fixed[s, ~isdelta] = u.ravel()
        fixed[s, isdelta] = (y >= 0).astype(np.uint8)
    return fixed


@pytest.mark.parametrize("name,nb", [("cfg2", 48), ("cfg3", 96)])
def test_lds_resident_lp_equals_dense_dictionary_lp(name, nb):
    """relaxation-only mode (BASELINE configs[1]: binaries fixed): the LDS-resident revised dual simplex (k_lp_lds) against the
    dense-dictionary kernel (opts.reserved bit 8) and the C oracle's LP on the same instances"""
    import orc
    import tighten_np
    wl = syn.make_workload(name, batch=nb)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost)
    q = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, reserved=256)
    fixed = _fixed_pattern(ag, wl, p, d, np.random.Generator(np.random.PCG64(9)), nb)
    a = p.solve(ag["x0"], ag["omega"], fixed_bin=fixed)
    b = q.solve(ag["x0"], ag["omega"], fixed_bin=fixed)
    assert np.array_equal(a["status"], b["status"]), (a["status"], b["status"])
    ok = a["status"] == 0
    assert ok.sum() >= nb // 2
    assert np.all(np.abs(a["obj"][ok] - b["obj"][ok]) <= 1e-9 * np.maximum(1.0, np.abs(b["obj"][ok])))
    assert np.all(a["nodes"] == 1)
    bins = np.where(p.is_bin)[0]
    assert np.array_equal(a["v"][:, bins][ok], fixed[ok].astype(float))
    sf = cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    for s in range(0, nb, 7):
        lb, ub = sf["lb"].copy(), sf["ub"].copy()
        lb[bins] = ub[bins] = fixed[s]
        ref = orc.solve_milp(cn.lin_cost(sf["cost"], ag["x0"][s], ag["omega"][s]), sf["G"], cn.rhs(sf["evo"], ag["x0"][s], ag["omega"][s]), lb, ub,
                             np.zeros_like(sf["is_bin"]), presolve=0, max_cuts=0, cut_rounds=0)
        r = cn.cost_const(sf["cost"]["const_terms"], ag["x0"][s], ag["omega"][s])
        assert gpu._lib.STATUS_NAMES[int(a["status"][s])] == ref["status"]
        if ref["status"] == "optimal":
            assert abs(a["obj"][s] - (ref["obj"] + r)) <= 1e-8 * max(1.0, abs(ref["obj"] + r))
    print("%s: LDS LP %.3f ms (%.0f/s, %d pivots), dense %.3f ms (%.0f/s, %d pivots)" % (
        name, a["stats"]["solve_ms"], nb / a["stats"]["solve_ms"] * 1e3, a["stats"]["pivots"], b["stats"]["solve_ms"], nb / b["stats"]["solve_ms"] * 1e3, b["stats"]["pivots"]))
    p.close(); q.close(); m.close()


def test_lds_lp_overflow_fallback_and_model_switching():
    """k_lp_lds edge cases: (i) instances whose working basis outgrows the LDS capacity (forced: opts.reserved bit 9 caps it at 24) are
    re-solved by the dense kernel and the batch still equals the all-dense result; (ii) several models interleaved in one batch
    (the workgroup reloads the Toeplitz blocks on every model change); (iii) an infeasible LP keeps its status"""
    wl = syn.make_workload("cfg3", batch=20, n_agents=3)
    d = wl["agents"][0]["dims"]
    N = wl["N_tilde"]
    m = gpu.GpuModel([a["mats"] for a in wl["agents"]], d)
    cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, wl["N_p"], N) for a in wl["agents"]])
    x0 = np.concatenate([a["x0"] for a in wl["agents"]])
    om = np.concatenate([a["omega"] for a in wl["agents"]])
    midx = np.repeat(np.arange(3), 20).astype(np.int32)
    order = np.random.Generator(np.random.PCG64(3)).permutation(60)
    x0, om, midx = x0[order], om[order], midx[order]
    ref_p = gpu.GpuProblem(m, wl["N_p"], N, cost, reserved=256)
    fixed = np.concatenate([_fixed_pattern(a, wl, ref_p, d, np.random.Generator(np.random.PCG64(20 + i)), 20) for i, a in enumerate(wl["agents"])])[order]
    ref = ref_p.solve(x0, om, midx, fixed_bin=fixed)
    for flags in (0, 512):
        p = gpu.GpuProblem(m, wl["N_p"], N, cost, reserved=flags)
        out = p.solve(x0, om, midx, fixed_bin=fixed)
        assert np.array_equal(out["status"], ref["status"]), flags
        ok = ref["status"] == 0
        assert ok.sum() >= 30
        assert np.all(np.abs(out["obj"][ok] - ref["obj"][ok]) <= 1e-9 * np.maximum(1.0, np.abs(ref["obj"][ok]))), flags
        if flags == 512:
            assert (out["pivots"] != ref["pivots"]).sum() < 60          # (the fall-back instances ran on the dense kernel: same pivot counts there)
        p.close()
    ref_p.close(); m.close()
    # (iii) an infeasible LP: hard bound 1 <= x <= ... contradiction (no soft slack), one binary input fixed
    import pyhybridcontrol_amd as phc
    hard = dict(A=[[1.0]], B1=[[0.0]], E=[[1.0], [-1.0]], F1=[[0.0], [0.0]], f5=[[1.0], [-2.0]])
    dims = dict(nx=1, nu=1, ndelta=0, nz=0, nmu=0, nomega=0, ny=1, nc=2, nu_l=1, nmu_l=0)
    gm = gpu.GpuModel([dict(hard, C=[[1.0]])], dims)
    for flags in (0, 256):
        gp = gpu.GpuProblem(gm, 7, 8, host.cost_from_atoms({"q_u": 1.0}, dims, 7, 8), reserved=flags)
        o = gp.solve(np.zeros((2, 1)), np.zeros((2, 0)), fixed_bin=np.zeros((2, 8), np.uint8))
        assert np.all(o["status"] == 1) and not np.isfinite(o["obj"]).any(), (flags, o["status"])
        gp.close()
    gm.close()


def test_launch_finish_on_two_streams_equals_resident_solves():
    """mld_solve_launch / mld_solve_finish on two problem handles with their own HIP streams (bench.py's timed loop: step k+1 is queued while
    step k runs) return, step by step, exactly what mld_solve_resident returns on one handle: objectives, statuses, inputs, node counts"""
    nb, K = 96, 5
    wl = syn.make_workload("cfg3", batch=nb)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
    rng = np.random.Generator(np.random.PCG64(77))
    X = np.stack([ag["x0"]] + [ag["x0"] + 0.5 * rng.standard_normal(ag["x0"].shape) for _ in range(K)])
    W = np.stack([ag["omega"]] + [ag["omega"] * (1.0 + 0.05 * rng.standard_normal(ag["omega"].shape)) for _ in range(K)])
    kw = dict(gap_rel=1e-3, max_nodes=400, max_pivots=40000)
    ref_p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, **kw)
    ref_p.upload(X[0], W[0]); ref_p.stage(X, W)
    ref = []
    for k in range(1, K + 1):
        ref_p.select(k); st = ref_p.solve_resident(); out = ref_p.download()
        ref.append((st["nodes"], st["pivots"], out["obj"].copy(), out["status"].copy(), out["v"].copy()))
    probs = [gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, **kw) for _ in range(2)]
    for p in probs:
        p.upload(X[0], W[0]); p.stage(X, W); p.use_stream()
    got = [None] * K

    def collect(p, k):
        st = p.finish(); out = p.download()
        got[k - 1] = (st["nodes"], st["pivots"], out["obj"].copy(), out["status"].copy(), out["v"].copy())

    for k in range(1, K + 1):
        p = probs[k % 2]
        if k > 2:
            collect(p, k - 2)
        p.select(k); p.launch()
    for k in (K - 1, K):
        collect(probs[k % 2], k)
    for k in range(K):
        assert ref[k][0] == got[k][0] and ref[k][1] == got[k][1], (k, ref[k][:2], got[k][:2])
        assert np.array_equal(ref[k][3], got[k][3]) and np.array_equal(ref[k][2], got[k][2]) and np.array_equal(ref[k][4], got[k][4])
    with pytest.raises(Exception):
        probs[0].finish()               # nothing in flight any more
    probs[0].launch()
    with pytest.raises(Exception):
        probs[0].launch()               # the previous launch has not been finished
    with pytest.raises(Exception):
        probs[0].download()             # ... and owns the buffers until it is
    with pytest.raises(Exception):
        probs[0].select(1)
    probs[0].finish()
    probs[0].download()
    for p in probs + [ref_p]:
        p.close()
    m.close()
