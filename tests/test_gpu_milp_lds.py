"""k_milp_lds -- the EXPERIMENTAL branch-and-cut on the LDS-resident formulation (opts.reserved bit 10) -- against the
dense-dictionary kernel (the default path) and against the HiGHS optima in tests/golden/.

The contract is the one of the default path (controllers/controller_base.py:509, :533-535): a returned point is feasible and its
objective is never below the optimum, a reported lower bound is never above it, OPTIMAL means within the requested gap.  Instances this
engine cannot finish (working basis larger than its LDS capacity, numerical trouble) come back from the dense kernel: the caller never
sees a status -1 (opts.reserved bit 11 makes them visible for the counting below).
"""
import os

import numpy as np
import pytest

import bench
from pyhybridcontrol_amd import gpu, host, synthetic as syn
from test_gpu_bench_parity import GDIR, _check_against_optimum
from test_gpu_solve import check_solution

pytestmark = pytest.mark.gpu

LDS, SHOW_FALLBACK, SMALL_BASIS = 1024, 2048, 512


def _one_agent(name, nb):
    wl = syn.make_workload(name, batch=nb)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    return wl, ag, d, m, host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])


def test_lds_branch_and_cut_equals_dense_kernel_at_zero_gap():
    """cfg2, gap 0: both engines prove the same optimum on every instance (objective 1e-6, binaries integral, rows satisfied), with cuts"""
    nb = 64
    wl, ag, d, m, cost = _one_agent("cfg2", nb)
    kw = dict(gap_rel=0.0, gap_abs=1e-9, max_nodes=20000, max_pivots=400000)
    a = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, **kw)
    b = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, reserved=LDS | SHOW_FALLBACK, **kw)
    ra, rb = a.solve(ag["x0"], ag["omega"]), b.solve(ag["x0"], ag["omega"])
    assert np.all(ra["status"] == 0) and np.all(rb["status"] == 0), (np.unique(ra["status"]), np.unique(rb["status"]))
    rel = np.abs(ra["obj"] - rb["obj"]) / np.maximum(1.0, np.abs(ra["obj"]))
    assert rel.max() <= 1e-6, rel.max()
    assert np.all(rb["lower_bound"] <= rb["obj"] + 1e-9) and np.all(rb["obj"] - rb["lower_bound"] <= 1e-6 * np.maximum(1.0, np.abs(rb["obj"])))
    bins = np.where(b.is_bin)[0]
    assert np.abs(rb["v"][:, bins] - np.rint(rb["v"][:, bins])).max() == 0.0
    for s in range(0, nb, 5):
        check_solution(ag, wl, s, rb["v"][s], rb["obj"][s])
    assert rb["stats"]["cuts"] > 0 and rb["stats"]["nodes"] >= nb
    print("cfg2 zero gap: dense %d nodes %d pivots %.1f ms | LDS %d nodes %d pivots %.1f ms | max rel diff %.2e" % (
        ra["stats"]["nodes"], ra["stats"]["pivots"], ra["stats"]["solve_ms"], rb["stats"]["nodes"], rb["stats"]["pivots"], rb["stats"]["solve_ms"], rel.max()))
    a.close(); b.close(); m.close()


def test_lds_branch_and_cut_against_highs_optimum_cfg3():
    """cfg3 at the exact contract (gap 1e-6) against the HiGHS optima; the fall-back to the dense kernel is part of the product path"""
    gold = np.load(os.path.join(GDIR, "solve_cfg3.npz"))
    nb = int(gold["n_scen"])
    wl, ag, d, m, cost = _one_agent("cfg3", nb)
    kw = dict(gap_rel=1e-6, max_nodes=20000, max_pivots=400000)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, reserved=LDS, **kw)
    out = p.solve(ag["x0"], ag["omega"])
    assert not np.any(out["status"] == -1)
    rel = _check_against_optimum(out, gold["obj"], 1e-6)
    proven = out["status"] == 0
    assert proven.mean() >= 0.93, proven.mean()
    assert np.abs(rel[proven]).max() <= 2e-6
    q = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, reserved=LDS | SHOW_FALLBACK, **kw)
    raw = q.solve(ag["x0"], ag["omega"])
    own = raw["status"] != -1
    print("cfg3 exact: proven %d of %d; solved by k_milp_lds itself %d, re-solved by the dense kernel %d" % (proven.sum(), nb, own.sum(), (~own).sum()))
    assert own.mean() >= 0.5
    assert np.all(np.abs(raw["obj"][own] - out["obj"][own]) == 0.0), "the same kernel on the same instance is bit-reproducible"
    p.close(); q.close(); m.close()


def test_lds_branch_and_cut_bench_workload_within_gap():
    """256 instances of the bench shard (64 models: the Toeplitz blocks are re-staged when the model changes), bench options"""
    n_scen = 4
    gold = np.load(os.path.join(GDIR, "solve_cfg4_bench.npz"))
    opt = gold["obj"][: n_scen * 64]
    agents, N_p, N_t, x0, om, midx = bench.make_shard(64, n_scen, 0)
    dd = agents[0]["dims"]
    model = gpu.GpuModel([a["mats"] for a in agents], dd)
    cost = host.stack_costs([host.cost_from_atoms(a["atoms"], dd, N_p, N_t) for a in agents])
    p = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-2, max_nodes=800, max_pivots=40000, reserved=LDS)
    out = p.solve(x0, om, midx)
    rel = _check_against_optimum(out, opt, 1e-2)
    assert (out["status"] == 0).mean() >= 0.97
    assert rel.max() <= 0.10
    p.close(); model.close()


def test_lds_branch_and_cut_overflow_falls_back_to_dense():
    """a working-basis capacity of 24 (opts.reserved bit 9) is too small for cfg2: the instances report -1 inside, and the caller gets the
    dense kernel's answer"""
    nb = 32
    wl, ag, d, m, cost = _one_agent("cfg2", nb)
    kw = dict(gap_rel=1e-4, max_nodes=4000, max_pivots=100000)
    ref = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, **kw).solve(ag["x0"], ag["omega"])
    raw = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, reserved=LDS | SMALL_BASIS | SHOW_FALLBACK, **kw).solve(ag["x0"], ag["omega"])
    out = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, reserved=LDS | SMALL_BASIS, **kw).solve(ag["x0"], ag["omega"])
    over = raw["status"] == -1
    assert over.sum() >= nb // 2, over.sum()
    assert not np.any(out["status"] == -1)
    assert np.array_equal(out["status"][over], ref["status"][over])
    assert np.all(out["obj"][over] == ref["obj"][over]), "re-solved instances are the dense kernel's results, bit for bit"
    assert np.all(np.abs(out["obj"] - ref["obj"]) <= 2e-4 * np.maximum(1.0, np.abs(ref["obj"])))
    m.close()


@pytest.mark.parametrize("seed", range(12))
def test_lds_branch_and_cut_on_random_mld_models_matches_highs(seed):
    """models that are NOT the tank clusters (tests/test_gpu_fuzz.py: random couplings, free auxiliaries, short horizon): the LDS engine's
    compact lag-block store, step buckets, cuts and search against scipy's HiGHS on the original rows; UNBOUNDED must be reported too"""
    from scipy.optimize import Bounds, LinearConstraint, milp
    import condense_np as cn
    from test_gpu_fuzz import random_mld
    mats, dims, atoms, rng = random_mld(seed)
    N_p, N, nb = 4, 5, 6
    x0 = rng.standard_normal((nb, dims["nx"]))
    om = rng.standard_normal((nb, N * dims["nomega"]))
    m = gpu.GpuModel([mats], dims)
    cost = host.cost_from_atoms(atoms, dims, N_p, N)
    raw_out = gpu.GpuProblem(m, N_p, N, cost, max_nodes=50000, max_pivots=400000, reserved=LDS | SHOW_FALLBACK).solve(x0, om)
    out = gpu.GpuProblem(m, N_p, N, cost, max_nodes=50000, max_pivots=400000, reserved=LDS).solve(x0, om)
    m.close()
    raw = cn.standard_form(mats, atoms, N_p, N, nu_l=dims["nu_l"])
    own = 0
    for s in range(nb):
        w = om[s] if dims["nomega"] else np.zeros(0)
        h, q = cn.rhs(raw["evo"], x0[s], w), cn.lin_cost(raw["cost"], x0[s], w)
        r = cn.cost_const(raw["cost"]["const_terms"], x0[s], w)
        ref = milp(q, constraints=LinearConstraint(raw["G"], -np.inf, h), integrality=raw["is_bin"].astype(int),
                   bounds=Bounds(raw["lb"], raw["ub"]), options=dict(mip_rel_gap=0.0))
        own += int(raw_out["status"][s] != -1)
        if ref.status == 3:
            assert out["status"][s] == 4, (seed, s, out["status"][s], raw_out["status"][s])
            continue
        assert ref.status == 0, (seed, s, ref.status)
        assert out["status"][s] == 0, (seed, s, out["status"][s], raw_out["status"][s], out["nodes"][s])
        assert abs(out["obj"][s] - (ref.fun + r)) <= 1e-6 * max(1.0, abs(ref.fun + r)), (seed, s, out["obj"][s], ref.fun + r, raw_out["status"][s])
        v = out["v"][s]
        bins = raw["is_bin"]
        assert np.all((v[bins] == 0) | (v[bins] == 1))
        assert np.all(raw["G"] @ v - h <= 1e-6 * np.maximum(1.0, np.abs(raw["G"]).max(axis=1)))
    print("seed %d: %d of %d instances finished by k_milp_lds itself" % (seed, own, nb))
