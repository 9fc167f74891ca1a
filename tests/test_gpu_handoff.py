"""Sub-tree hand-off (round 3): the open nodes of a search that stopped at its node limit are read off its depth-first stack
(mld_download_open_nodes) and solved as instances of the next batch under the parent's incumbent as cutoff (mld_set_cutoffs), driven by
GpuProblem.solve_handoff.  The union of the open nodes is exactly what the stopped search had left, so the merged answer must be the answer
of an unlimited search: checked against the plain solve, the oracle and the committed HiGHS optima."""
import os

import numpy as np
import pytest

import bench
import condense_np as cn
import orc
import tighten_np
from pyhybridcontrol_amd import MldGpuError, gpu, host, synthetic as syn
from test_gpu_bench_parity import _check_against_optimum

pytestmark = pytest.mark.gpu

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cfg2(batch, **opts):
    wl = syn.make_workload("cfg2", batch=batch)
    ag = wl["agents"][0]
    d = ag["dims"]
    m = gpu.GpuModel([ag["mats"]], d)
    p = gpu.GpuProblem(m, wl["N_p"], wl["N_tilde"], host.cost_from_atoms(ag["atoms"], d, wl["N_p"], wl["N_tilde"]), **opts)
    return wl, ag, m, p


def test_cutoff_semantics():
    """under a cutoff only better points count: a cutoff above the optimum changes nothing, one below it ends INFEASIBLE ("nothing better")"""
    wl, ag, m, p = _cfg2(16, gap_rel=0.0, max_nodes=100000)
    ref = p.solve(ag["x0"], ag["omega"])
    assert np.all(ref["status"] == 0)
    p.upload(ag["x0"], ag["omega"])
    p.set_cutoffs(ref["obj"] + 1.0)
    p.solve_resident(); hi = p.download()
    assert np.all(hi["status"] == 0) and np.allclose(hi["obj"], ref["obj"], rtol=1e-9, atol=1e-9)
    p.upload(ag["x0"], ag["omega"])
    p.set_cutoffs(ref["obj"] - 1e-6 * np.maximum(1.0, np.abs(ref["obj"])))
    p.solve_resident(); lo = p.download()
    assert np.all(lo["status"] == 1) and not np.any(np.isfinite(lo["obj"]))
    p.upload(ag["x0"], ag["omega"])                       # an upload clears the cutoffs
    p.solve_resident(); again = p.download()
    assert np.array_equal(again["obj"], ref["obj"])
    with pytest.raises(MldGpuError):
        p.open_nodes()                                    # recording was not enabled
    p.close(); m.close()


def test_handoff_with_a_tiny_first_pass_equals_the_unlimited_search():
    """first pass of 3 nodes per instance, 12 per open node afterwards: the instances that need a tree are handed off, several rounds deep; the merged
    result is the exact optimum"""
    wl, ag, m, p = _cfg2(48, gap_rel=0.0, max_nodes=100000, cut_rounds=1)       # (one cut round: the full loop closes all but one of these instances at the root)
    ref = p.solve(ag["x0"], ag["omega"])
    assert np.all(ref["status"] == 0)
    out = p.solve_handoff(ag["x0"], ag["omega"], first_nodes=3, sub_nodes=12, rounds=30, max_open=None)
    print("handoff:", out["handoff"])
    assert out["handoff"]["handed_off"] >= 3 and len(out["handoff"]["rounds"]) >= 2
    assert np.all(out["status"] == 0), np.unique(out["status"], return_counts=True)
    assert np.allclose(out["obj"], ref["obj"], rtol=1e-9, atol=1e-9)
    assert np.all(out["lower_bound"] <= out["obj"] + 1e-9) and np.all(out["lower_bound"] >= ref["obj"] - 1e-6 * np.maximum(1.0, np.abs(ref["obj"])))
    # the returned points are the incumbents of those objectives: binaries exact, original rows satisfied (oracle standard form)
    d = ag["dims"]
    sf = cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    for s in range(0, 48, 5):
        v = out["v"][s]
        assert np.all((v[sf["is_bin"]] == 0) | (v[sf["is_bin"]] == 1))
        h = cn.rhs(sf["evo"], ag["x0"][s], ag["omega"][s])
        assert np.all(sf["G"] @ v - h <= 1e-6 * np.maximum(1.0, np.abs(sf["G"]).max(axis=1)))
        q, r = cn.lin_cost(sf["cost"], ag["x0"][s], ag["omega"][s]), cn.cost_const(sf["cost"]["const_terms"], ag["x0"][s], ag["omega"][s])
        assert abs(q @ v + r - out["obj"][s]) <= 1e-7 * max(1.0, abs(out["obj"][s]))
    assert p.opts.max_nodes == 100000, "the problem's own limits are restored"
    p.close(); m.close()


def test_handoff_on_the_bench_shard_at_the_exact_contract():
    """first 256 instances of the bench shard at gap 1e-6: a first pass of 500 nodes, open nodes re-queued with 2000 nodes each -- every instance
    the one-workgroup-per-instance search proves with 20 000 nodes is proven, every proven objective equals the HiGHS optimum"""
    agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 4, 0)
    d = agents[0]["dims"]
    model = gpu.GpuModel([a["mats"] for a in agents], d)
    cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
    opt = np.load(os.path.join(GDIR, "solve_cfg4_bench.npz"))["obj"][:256]
    prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-6, max_nodes=20000, max_pivots=400000)
    plain = prob.solve(x0, om, midx)
    out = prob.solve_handoff(x0, om, midx, first_nodes=500, sub_nodes=2000, rounds=4)
    print("bench handoff:", out["handoff"], "proven plain %d handoff %d" % ((plain["status"] == 0).sum(), (out["status"] == 0).sum()))
    rel = _check_against_optimum(out, opt, 1e-6)
    proven = out["status"] == 0
    assert np.abs(rel[proven]).max() <= 2e-6
    assert proven.sum() >= (plain["status"] == 0).sum()
    assert np.all(out["obj"] <= plain["obj"] + 1e-9 * np.maximum(1.0, np.abs(plain["obj"]))), "the hand-off never ends with a worse incumbent"
    prob.close(); model.close()


def test_in_kernel_handoff_with_a_tiny_first_pass_equals_the_unlimited_search():
    """the same exactness check with the hand-off INSIDE the launch (mld_set_handoff): 3 nodes per instance, 12 per item, items split again up to
    ten generations deep; the merged result per instance is the exact optimum, and a second run returns the same bits"""
    wl, ag, m, p = _cfg2(48, gap_rel=0.0, max_nodes=100000, cut_rounds=1)
    ref = p.solve(ag["x0"], ag["omega"])
    assert np.all(ref["status"] == 0)
    out = p.solve_handoff_device(ag["x0"], ag["omega"], first_nodes=3, sub_nodes=12, max_gen=8, max_children=64, max_tree=100000, room_factor=64.0)
    print("in-kernel handoff:", out["handoff"], np.unique(out["status"], return_counts=True))
    assert out["handoff"]["items"] >= 3
    assert np.all(out["status"] == 0), np.unique(out["status"], return_counts=True)
    assert np.allclose(out["obj"], ref["obj"], rtol=1e-9, atol=1e-9)
    assert np.all(out["lower_bound"] <= out["obj"] + 1e-9) and np.all(out["lower_bound"] >= ref["obj"] - 1e-6 * np.maximum(1.0, np.abs(ref["obj"])))
    d = ag["dims"]
    sf = cn.standard_form(ag["mats"], ag["atoms"], wl["N_p"], wl["N_tilde"], nu_l=d["nu_l"])
    for s in range(48):
        v = out["v"][s]
        assert np.all((v[sf["is_bin"]] == 0) | (v[sf["is_bin"]] == 1))
        h = cn.rhs(sf["evo"], ag["x0"][s], ag["omega"][s])
        assert np.all(sf["G"] @ v - h <= 1e-6 * np.maximum(1.0, np.abs(sf["G"]).max(axis=1)))
        q, r = cn.lin_cost(sf["cost"], ag["x0"][s], ag["omega"][s]), cn.cost_const(sf["cost"]["const_terms"], ag["x0"][s], ag["omega"][s])
        assert abs(q @ v + r - out["obj"][s]) <= 1e-7 * max(1.0, abs(out["obj"][s]))
    again = p.solve_handoff_device(ag["x0"], ag["omega"], first_nodes=3, sub_nodes=12, max_gen=8, max_children=64, max_tree=100000, room_factor=64.0)
    assert np.array_equal(again["obj"], out["obj"]) and np.array_equal(again["v"], out["v"]) and np.array_equal(again["status"], out["status"]), "reproducible whatever the queue order"
    assert p.opts.max_nodes == 100000
    plain = p.solve(ag["x0"], ag["omega"])                 # the switch is off again: a plain solve is the plain solve
    assert np.array_equal(plain["obj"], ref["obj"])
    p.close(); m.close()


def test_in_kernel_handoff_on_the_bench_shard_at_the_exact_contract_and_for_one_instance():
    """256 bench instances at gap 1e-6 in ONE launch: at least what the one-workgroup-per-instance search proves, every proven objective at the HiGHS
    optimum; and batch 1 (what MpcController.solve does): the instance's tree spreads over the idle workgroups"""
    agents, N_p, N_t, x0, om, midx = bench.make_shard(64, 4, 0)
    d = agents[0]["dims"]
    model = gpu.GpuModel([a["mats"] for a in agents], d)
    cost = host.stack_costs([host.cost_from_atoms(a["atoms"], d, N_p, N_t) for a in agents])
    opt = np.load(os.path.join(GDIR, "solve_cfg4_bench.npz"))["obj"][:256]
    prob = gpu.GpuProblem(model, N_p, N_t, cost, gap_rel=1e-6, max_nodes=20000, max_pivots=400000)
    plain = prob.solve(x0, om, midx)
    out = prob.solve_handoff_device(x0, om, midx, first_nodes=300, sub_nodes=200, max_gen=8)
    print("bench in-kernel handoff:", out["handoff"], "proven plain %d in-kernel %d  kernel ms plain %.0f in-kernel %.0f" %
          ((plain["status"] == 0).sum(), (out["status"] == 0).sum(), plain["stats"]["solve_ms"], out["stats"]["solve_ms"]))
    rel = _check_against_optimum(out, opt, 1e-6)
    proven = out["status"] == 0
    assert np.abs(rel[proven]).max() <= 2e-6
    assert proven.sum() >= (plain["status"] == 0).sum() - 1
    hard = int(np.argmax(plain["nodes"]))
    one_plain = prob.solve(x0[hard:hard + 1], om[hard:hard + 1], midx[hard:hard + 1])
    one = prob.solve_handoff_device(x0[hard:hard + 1], om[hard:hard + 1], midx[hard:hard + 1], first_nodes=100, sub_nodes=200, max_gen=8)
    print("one instance (%d nodes plain): kernel ms plain %.1f in-kernel hand-off %.1f items %d status %d" %
          (plain["nodes"][hard], one_plain["stats"]["solve_ms"], one["stats"]["solve_ms"], one["handoff"]["items"], one["status"][0]))
    if one["status"][0] == 0:
        assert abs(one["obj"][0] - opt[hard]) <= 2e-6 * max(1.0, abs(opt[hard]))
    prob.close(); model.close()
