"""Round 4: host-API leftovers of the reference's controller and the handle-state fixes of ADVICE r3, on the GPU.

* build(with_std_constraints=False) and set_constraints(std_evo_constaints=[...]) (controllers/mpc_controller.py:76-101,
  controllers/controller_base.py:457-475): the standard block is replaced by the given blocks (mld_set_std_block);
* set_objective(other_objectives=[...]) (mpc_controller.py:63-74): further atom sets added to the standard objective;
* mld_advance_batch2 refuses a second advance without a solve in between; mld_warm_start_from_previous needs a finished solve for every shift;
  mld_problem_set_opts changes nothing when it fails.
"""
import numpy as np
import pytest

import pyhybridcontrol_amd as phc
from pyhybridcontrol_amd import MldGpuError, gpu, host, synthetic as syn

pytestmark = pytest.mark.gpu

PRICE = np.array([1, 3, 3, 1, 1.0])
W1 = [.004, .012, 0, .009, .002]
W2 = [.010, .000, .015, .002, .011]


def _dewh():
    return phc.MldModel(A=[[0.9970371127900564]], B1=[[4.298192277481107]], B4=[[-179.73320827515]], b5=[[0.07407218024859108]],
                        E=[[1], [-1]], F1=[[0], [0]], Psi=[[-1, 0], [0, -1]], f5=[[65.0], [-50.0]], nu_l=1, ts=900)


def _ctrl(**atoms):
    c = phc.MpcController(_dewh(), N_p=4)
    c.set_std_obj_atoms(**(atoms or dict(q_u=(PRICE * 0.75).reshape(-1, 1), q_mu=[90.0, 90.0])))
    return c


def test_build_without_the_standard_block_and_custom_standard_blocks():
    ref = _ctrl(); ref.build()
    want = ref.solve(0, x_k=[50.3], omega_tilde_k=W2)                   # the problem whose only constraints are the W2 block
    v_want = ref.v_N_tilde.copy()
    c = _ctrl(); c.build()
    blk = c.gen_evo_constraints(x_k=None, omega_tilde_k=np.array(W2).reshape(-1, 1))
    c.set_constraints(other_constraints=[blk])
    c.build(with_std_constraints=False)                                  # standard block (W1) out: only the W2 block constrains
    got = c.solve(0, x_k=[50.3], omega_tilde_k=W1)
    assert abs(got - want) <= 1e-9 * max(1.0, abs(want)) and np.array_equal(c.v_N_tilde, v_want)
    both = _ctrl(); both.build()
    both.set_constraints(other_constraints=[blk]); both.build()          # standard block (W1) AND the W2 block: at least as expensive as either
    tight = both.solve(0, x_k=[50.3], omega_tilde_k=W1)
    assert tight >= want - 1e-9
    c2 = _ctrl(); c2.build()
    c2.set_constraints(std_evo_constaints=[blk]); c2._build_required = False       # the same through set_constraints(std_evo_constaints=[...])
    c2._problem.set_std_block(False)
    assert abs(c2.solve(0, x_k=[50.3], omega_tilde_k=W1) - want) <= 1e-9 * max(1.0, abs(want))
    free = _ctrl(); free.build(with_std_constraints=False)               # no block at all: nothing constrains, nothing is heated, no slack is paid
    assert abs(free.solve(0, x_k=[50.3], omega_tilde_k=W1)) <= 1e-12 and not free.v_N_tilde.any()
    free.build()                                                         # and back: the standard block is part of the problem again
    back = free.solve(0, x_k=[50.3], omega_tilde_k=W2)
    assert abs(back - want) <= 1e-9 * max(1.0, abs(want))


def test_other_objectives_are_added_to_the_standard_objective():
    both = _ctrl(q_u=(PRICE * 0.75).reshape(-1, 1), q_mu=[90.0, 90.0]); both.build()
    want = both.solve(0, x_k=[50.3], omega_tilde_k=W1)
    c = _ctrl(q_u=(PRICE * 0.5).reshape(-1, 1), q_mu=[90.0, 90.0])
    c.set_objective(other_objectives=[dict(q_u=(PRICE * 0.25).reshape(-1, 1))])
    c.build()
    got = c.solve(0, x_k=[50.3], omega_tilde_k=W1)
    assert abs(got - want) <= 1e-9 * max(1.0, abs(want)) and np.array_equal(c.v_N_tilde, both.v_N_tilde)
    c.set_objective(other_objectives=None)
    c.build()
    assert c.solve(0, x_k=[50.3], omega_tilde_k=W1) < want - 1e-6        # the cheaper standard objective alone
    with pytest.raises(NotImplementedError):
        c.set_objective(other_objectives=[dict(q_L1_du=1.0)])


def test_handle_state_after_advance_and_failed_set_opts():
    wl = syn.make_workload("cfg2", batch=8)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), gap_rel=1e-2, max_nodes=200)
    p.upload(ag["x0"], ag["omega"])
    with pytest.raises(MldGpuError):
        p.warm_start_from_previous(1)                                    # no finished solve of this batch: no plan to move on (any shift)
    p.solve_resident()
    p.advance()
    with pytest.raises(MldGpuError):
        p.advance()                                                      # the plan has been applied: a second advance needs the next solve
    p.warm_start_from_previous(1)
    p.solve_resident()
    p.advance()                                                          # fine again
    before = {k: getattr(p.opts, k) for k in ("gap_rel", "max_nodes", "max_pivots", "reserved", "flags")}
    with pytest.raises((MldGpuError, TypeError)):
        p.set_opts(max_nodes=7, flags=1)                                 # flags cannot change: NOTHING of the call may stick
    assert {k: getattr(p.opts, k) for k in before} == before
    p.solve_resident()
    st = p.download()
    assert np.all(st["nodes"] <= 200 + 3 * p.n_bin + 12), "the failed call must not have installed its node limit on the device"
    p.close(); m.close()


def test_cfg2_relaxation_only_at_its_stated_batch_of_256():
    """BASELINE configs[1] as stated: single agent, N = 24, batch 256, QP-relaxation kernel only (binaries fixed).  The LDS-resident revised simplex
    (k_lp_lds) against the dense-dictionary kernel on all 256 instances and against the C oracle's LP on every eighth one; and the same batch with
    the MIQP variant's quadratic cost (dense kernel, simplicial decomposition) against the oracle's QP on every 32nd."""
    from oracle import condense_np as cn, orc, tighten_np
    from test_gpu_loop import _fixed_pattern
    nb = 256
    for quad in (False, True):
        wl = syn.make_workload("cfg2", batch=nb, quadratic=quad)
        ag = wl["agents"][0]
        d = ag["dims"]
        m = gpu.GpuModel([ag["mats"]], d)
        cost = host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"])
        p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost)
        fixed = _fixed_pattern(ag, wl, p, d, np.random.Generator(np.random.PCG64(11)), nb)
        a = p.solve(ag["x0"], ag["omega"], fixed_bin=fixed)
        assert np.all(a["nodes"] == 1)
        ok = a["status"] == 0
        assert ok.sum() >= nb // 2
        if not quad:
            q = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], cost, reserved=256)          # the dense-dictionary kernel on the same LPs
            b = q.solve(ag["x0"], ag["omega"], fixed_bin=fixed)
            assert np.array_equal(a["status"], b["status"])
            assert np.all(np.abs(a["obj"][ok] - b["obj"][ok]) <= 1e-9 * np.maximum(1.0, np.abs(b["obj"][ok])))
            q.close()
        sf = cn.standard_form(tighten_np.tighten(ag["mats"], d, nu_l=d["nu_l"]), ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
        bins = np.where(sf["is_bin"])[0]
        for s in range(0, nb, 32 if quad else 8):
            lb, ub = sf["lb"].copy(), sf["ub"].copy()
            lb[bins] = ub[bins] = fixed[s]
            h, qv, r = cn.rhs(sf["evo"], ag["x0"][s], ag["omega"][s]), cn.lin_cost(sf["cost"], ag["x0"][s], ag["omega"][s]), cn.cost_const(sf["cost"]["const_terms"], ag["x0"][s], ag["omega"][s])
            isb0 = np.zeros_like(sf["is_bin"])
            ref = (orc.solve_miqp(sf["cost"]["P"], qv, sf["G"], h, lb, ub, isb0, max_nodes=10, presolve=0) if quad
                   else orc.solve_milp(qv, sf["G"], h, lb, ub, isb0, max_nodes=10, presolve=0))
            assert (ref["status"] == "optimal") == bool(ok[s]), (quad, s, ref["status"], a["status"][s])
            if ok[s]:
                assert abs(a["obj"][s] - (ref["obj"] + r)) <= 1e-6 * max(1.0, abs(ref["obj"] + r)), (quad, s, a["obj"][s], ref["obj"] + r)
        p.close(); m.close()
