"""Parity on models that are NOT the tank clusters: seeded random MLD systems (binary inputs and deltas, continuous auxiliaries, every
row soft) condensed over a short horizon and solved exactly on the GPU, against scipy's HiGHS on the ORIGINAL rows and against the C
oracle -- branching rule, cuts, tightening and the dual simplex guards must not depend on the example's structure."""
import numpy as np
import pytest
from scipy.optimize import Bounds, LinearConstraint, milp

import condense_np as cn
import orc
import tighten_np
from pyhybridcontrol_amd import gpu, host

pytestmark = pytest.mark.gpu


def random_mld(seed):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    nx, nu, nd, nz = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(0, 3)), int(rng.integers(0, 3))
    nw, ny, nc = int(rng.integers(0, 3)), int(rng.integers(1, 3)), int(rng.integers(3, 7))
    nmu = nc
    A = 0.8 * rng.standard_normal((nx, nx)) / max(1, nx) ** 0.5
    m = dict(A=A, B1=rng.standard_normal((nx, nu)), B2=rng.standard_normal((nx, nd)), B3=0.5 * rng.standard_normal((nx, nz)),
             B4=rng.standard_normal((nx, nw)), b5=0.1 * rng.standard_normal((nx, 1)),
             C=rng.standard_normal((ny, nx)), D1=rng.standard_normal((ny, nu)), D2=np.zeros((ny, nd)), D3=np.zeros((ny, nz)),
             D4=np.zeros((ny, nw)), d5=np.zeros((ny, 1)),
             E=rng.standard_normal((nc, nx)), F1=3.0 * rng.standard_normal((nc, nu)), F2=5.0 * rng.standard_normal((nc, nd)),
             F3=rng.standard_normal((nc, nz)), F4=0.3 * rng.standard_normal((nc, nw)), f5=1.0 + rng.random((nc, 1)),
             G=0.5 * rng.standard_normal((nc, ny)), Psi=-np.eye(nc))
    dims = dict(nx=nx, nu=nu, ndelta=nd, nz=nz, nmu=nmu, nomega=nw, ny=ny, nc=nc, nu_l=nu, nmu_l=0)
    atoms = {"q_mu": 5.0 + 10.0 * rng.random(nmu), "q_u": rng.standard_normal(nu)}
    if nd:
        atoms["q_delta"] = rng.standard_normal(nd)
    return m, dims, atoms, rng


@pytest.mark.parametrize("seed", range(12))
def test_random_mld_models_match_highs_and_oracle(seed):
    mats, dims, atoms, rng = random_mld(seed)
    N_p, N = 4, 5
    nb = 6
    x0 = rng.standard_normal((nb, dims["nx"]))
    om = rng.standard_normal((nb, N * dims["nomega"]))
    m = gpu.GpuModel([mats], dims)
    p = gpu.GpuProblem(m, N_p, N, host.cost_from_atoms(atoms, dims, N_p, N), max_nodes=50000, max_pivots=400000)
    out = p.solve(x0, om)
    p.close(); m.close()
    raw = cn.standard_form(mats, atoms, N_p, N, nu_l=dims["nu_l"])
    tight = cn.standard_form(tighten_np.tighten(mats, dims, nu_l=dims["nu_l"]), atoms, N_p, N, nu_l=dims["nu_l"])
    for s in range(nb):
        w = om[s] if dims["nomega"] else np.zeros(0)
        h, q = cn.rhs(raw["evo"], x0[s], w), cn.lin_cost(raw["cost"], x0[s], w)
        r = cn.cost_const(raw["cost"]["const_terms"], x0[s], w)
        ref = milp(q, constraints=LinearConstraint(raw["G"], -np.inf, h), integrality=raw["is_bin"].astype(int),
                   bounds=Bounds(raw["lb"], raw["ub"]), options=dict(mip_rel_gap=0.0))
        if ref.status == 3:          # HiGHS: unbounded -- the GPU must say so too (status 4), never return a finite "optimum"
            assert out["status"][s] == 4, (seed, s, out["status"][s])
            continue
        assert ref.status == 0, (seed, s, ref.status)
        assert out["status"][s] == 0, (seed, s, out["status"][s], out["nodes"][s])
        assert abs(out["obj"][s] - (ref.fun + r)) <= 1e-6 * max(1.0, abs(ref.fun + r)), (seed, s, out["obj"][s], ref.fun + r)
        v = out["v"][s]
        bins = raw["is_bin"]
        assert np.all((v[bins] == 0) | (v[bins] == 1))
        assert np.all(raw["G"] @ v - h <= 1e-6 * np.maximum(1.0, np.abs(raw["G"]).max(axis=1)))
        o = orc.solve_milp(cn.lin_cost(tight["cost"], x0[s], w), tight["G"], cn.rhs(tight["evo"], x0[s], w), tight["lb"], tight["ub"], tight["is_bin"],
                           max_nodes=50000, presolve=0)
        assert o["status"] == "optimal" and abs(o["obj"] + r - out["obj"][s]) <= 1e-6 * max(1.0, abs(out["obj"][s])), (seed, s, o["status"], o["obj"] + r, out["obj"][s])


@pytest.mark.parametrize("seed", range(12))
def test_random_mld_relaxations_on_the_lds_kernel_match_highs(seed):
    """every binary fixed at a random value -> LPs on k_lp_lds (the LDS-resident revised simplex) for models that are not tank
    clusters, N_tilde = 8; against HiGHS's LP on the original rows and against the dense-dictionary kernel"""
    from scipy.optimize import linprog
    mats, dims, atoms, rng = random_mld(seed)
    N_p, N = 7, 8
    nb = 10
    x0 = rng.standard_normal((nb, dims["nx"]))
    om = rng.standard_normal((nb, N * dims["nomega"]))
    m = gpu.GpuModel([mats], dims)
    cost = host.cost_from_atoms(atoms, dims, N_p, N)
    p = gpu.GpuProblem(m, N_p, N, cost)
    q_dense = gpu.GpuProblem(m, N_p, N, cost, reserved=256)
    fixed = (rng.random((nb, p.n_bin)) < 0.5).astype(np.uint8)
    a = p.solve(x0, om, fixed_bin=fixed)
    b = q_dense.solve(x0, om, fixed_bin=fixed)
    p.close(); q_dense.close(); m.close()
    assert np.array_equal(a["status"], b["status"]), (seed, a["status"], b["status"])
    raw = cn.standard_form(mats, atoms, N_p, N, nu_l=dims["nu_l"])
    bins = np.where(raw["is_bin"])[0]
    for s in range(nb):
        w = om[s] if dims["nomega"] else np.zeros(0)
        h, q = cn.rhs(raw["evo"], x0[s], w), cn.lin_cost(raw["cost"], x0[s], w)
        r = cn.cost_const(raw["cost"]["const_terms"], x0[s], w)
        lb, ub = raw["lb"].copy(), raw["ub"].copy()
        lb[bins] = ub[bins] = fixed[s]
        ref = linprog(q, A_ub=raw["G"], b_ub=h, bounds=np.stack([lb, ub], axis=1), method="highs")
        if ref.status == 3:
            assert a["status"][s] == 4, (seed, s, a["status"][s])
            continue
        assert ref.status == 0 and a["status"][s] == 0, (seed, s, ref.status, a["status"][s])
        assert abs(a["obj"][s] - (ref.fun + r)) <= 1e-7 * max(1.0, abs(ref.fun + r)), (seed, s, a["obj"][s], ref.fun + r)
        assert abs(a["obj"][s] - b["obj"][s]) <= 2e-8 * max(1.0, abs(b["obj"][s]))          # (both stop at a primal tolerance of 1e-8 in their own scaling)
