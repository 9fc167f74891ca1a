"""ORACLE (test infrastructure): ctypes loader for oracle/_build/libmldoracle.so (mld_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libmldoracle.so")

STATUS = {0: "optimal", 1: "infeasible", 2: "node_limit", 3: "numerical", 4: "unbounded"}
MAT_NAMES = ("A", "B1", "B2", "B3", "B4", "b5", "C", "D1", "D2", "D3", "D4", "d5",
             "E", "F1", "F2", "F3", "F4", "f5", "G", "Psi")


class Dims(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("nx", "nu", "ndelta", "nz", "nmu", "nomega", "ny", "nc", "nu_l", "nmu_l")]


class Opts(C.Structure):
    _fields_ = [("gap_abs", C.c_double), ("gap_rel", C.c_double), ("max_nodes", C.c_int), ("cut_rounds", C.c_int),
                ("cuts_per_round", C.c_int), ("max_cuts", C.c_int), ("max_pivots", C.c_int), ("presolve", C.c_int),
                ("mir_per_round", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("nodes", C.c_int), ("pivots", C.c_int), ("cuts", C.c_int), ("refactors", C.c_int),
                ("status", C.c_int), ("root_lp", C.c_double), ("root_bound", C.c_double),
                ("lower_bound", C.c_double), ("work", C.c_double), ("bland", C.c_double), ("rebuilds", C.c_double), ("phase_work", C.c_double * 6), ("flips", C.c_double)]


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "mld_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_solve_milp.restype = C.c_int
        _lib.orc_enumerate_milp.restype = C.c_int
        _lib.orc_condense.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def make_opts(gap_abs=1e-9, gap_rel=0.0, max_nodes=100000, cut_rounds=10, cuts_per_round=80, max_cuts=300,
              max_pivots=0, presolve=1, mir_per_round=20):
    return Opts(gap_abs, gap_rel, max_nodes, cut_rounds, cuts_per_round, max_cuts, max_pivots, presolve, mir_per_round)


def solve_milp(q, G, h, lb, ub, is_bin, x_start=None, **kw):
    """x_start: MIP start (n values; its binaries, rounded, become the initial incumbent when feasible) -- orc_solve_miqp_start"""
    if x_start is not None:
        return _solve_start(None, q, G, h, lb, ub, is_bin, x_start, **kw)
    q = np.ascontiguousarray(q, np.float64)
    G = np.ascontiguousarray(G, np.float64)
    h = np.ascontiguousarray(h, np.float64)
    lb = np.ascontiguousarray(lb, np.float64)
    ub = np.ascontiguousarray(ub, np.float64)
    ib = np.ascontiguousarray(is_bin, np.uint8)
    m, n = G.shape
    x = np.zeros(n)
    obj = C.c_double()
    st = Stats()
    o = make_opts(**kw)
    s = lib().orc_solve_milp(n, m, _p(q), _p(G), _p(h), _p(lb), _p(ub), ib.ctypes.data_as(C.POINTER(C.c_ubyte)),
                             C.byref(o), _p(x), C.byref(obj), C.byref(st))
    return dict(status=STATUS[s], obj=obj.value, x=x if (np.isfinite(obj.value) or s == 4) else None, nodes=st.nodes,
                pivots=st.pivots, cuts=st.cuts, refactors=st.refactors, root_lp=st.root_lp,
                root_bound=st.root_bound, lower_bound=st.lower_bound, work=st.work, bland=st.bland, rebuilds=st.rebuilds, phase_work=list(st.phase_work), flips=st.flips)


def _solve_start(P, q, G, h, lb, ub, is_bin, x_start, **kw):
    q = np.ascontiguousarray(q, np.float64)
    G = np.ascontiguousarray(G, np.float64)
    h = np.ascontiguousarray(h, np.float64)
    lb = np.ascontiguousarray(lb, np.float64)
    ub = np.ascontiguousarray(ub, np.float64)
    ib = np.ascontiguousarray(is_bin, np.uint8)
    xs = np.ascontiguousarray(x_start, np.float64)
    Pc = np.ascontiguousarray(P, np.float64) if P is not None else None
    m, n = G.shape
    x = np.zeros(n)
    obj = C.c_double()
    st = Stats()
    o = make_opts(**kw)
    lib().orc_solve_miqp_start.restype = C.c_int
    s = lib().orc_solve_miqp_start(n, m, _p(Pc), _p(q), _p(G), _p(h), _p(lb), _p(ub), ib.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                   C.byref(o), _p(xs), _p(x), C.byref(obj), C.byref(st))
    return dict(status=STATUS[s], obj=obj.value, x=x if (np.isfinite(obj.value) or s == 4) else None, nodes=st.nodes,
                pivots=st.pivots, cuts=st.cuts, refactors=st.refactors, root_lp=st.root_lp,
                root_bound=st.root_bound, lower_bound=st.lower_bound, work=st.work, bland=st.bland, rebuilds=st.rebuilds, phase_work=list(st.phase_work), flips=st.flips)


def solve_milp_batch(qs, Gs, hs, lb, ub, is_bin, threads=0, **kw):
    """independent instances (q_i, G_i, h_i) over `threads` host cores (0 = all) with OpenMP: orc_solve_milp_batch.
    Gs may repeat the same array object for instances that share a model.  Returns (dict of arrays, threads used)."""
    n_inst = len(qs)
    qs = [np.ascontiguousarray(a, np.float64) for a in qs]
    hs = [np.ascontiguousarray(a, np.float64) for a in hs]
    keep = {}
    Gc = []
    for a in Gs:
        if id(a) not in keep:
            keep[id(a)] = np.ascontiguousarray(a, np.float64)
        Gc.append(keep[id(a)])
    m, n = Gc[0].shape
    lb = np.ascontiguousarray(lb, np.float64)
    ub = np.ascontiguousarray(ub, np.float64)
    ib = np.ascontiguousarray(is_bin, np.uint8)
    PP = C.POINTER(C.c_double) * n_inst
    qp, Gp, hp = PP(*[_p(a) for a in qs]), PP(*[_p(a) for a in Gc]), PP(*[_p(a) for a in hs])
    obj, lbd = np.zeros(n_inst), np.zeros(n_inst)
    st, nd, pv = np.zeros(n_inst, np.int32), np.zeros(n_inst, np.int32), np.zeros(n_inst, np.int32)
    o = make_opts(**kw)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    lib().orc_solve_milp_batch.restype = C.c_int
    used = lib().orc_solve_milp_batch(n_inst, n, m, qp, Gp, hp, _p(lb), _p(ub), ib.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(o),
                                      int(threads), _p(obj), ip(st), ip(nd), ip(pv), _p(lbd))
    return dict(obj=obj, status=st, nodes=nd, pivots=pv, lower_bound=lbd), used


def lp_revised(q, G, h, lb, ub, kcap=128, max_pivots=0):
    """the LP min q'x, Gx <= h, lb <= x <= ub by the revised dual simplex on the working basis (orc_lp_revised): the CPU
    restatement of the LDS-resident kernel's formulation.  Returns status, obj, x, pivots, refreshes, k (final basis size)."""
    q = np.ascontiguousarray(q, np.float64)
    G = np.ascontiguousarray(G, np.float64)
    h = np.ascontiguousarray(h, np.float64)
    lb = np.ascontiguousarray(lb, np.float64)
    ub = np.ascontiguousarray(ub, np.float64)
    m, n = G.shape
    x = np.zeros(n)
    obj = C.c_double()
    st = Stats()
    lib().orc_lp_revised.restype = C.c_int
    s = lib().orc_lp_revised(n, m, _p(q), _p(G), _p(h), _p(lb), _p(ub), int(kcap), C.c_long(int(max_pivots)), _p(x), C.byref(obj), C.byref(st))
    return dict(status=STATUS[s], obj=obj.value, x=x, pivots=st.pivots, refreshes=st.refactors, k=st.nodes)


def solve_miqp(P, q, G, h, lb, ub, is_bin, **kw):
    """min 1/2 x'Px + q'x over the mixed-integer polytope (P symmetric PSD)"""
    P = np.ascontiguousarray(P, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    G = np.ascontiguousarray(G, np.float64)
    h = np.ascontiguousarray(h, np.float64)
    lb = np.ascontiguousarray(lb, np.float64)
    ub = np.ascontiguousarray(ub, np.float64)
    ib = np.ascontiguousarray(is_bin, np.uint8)
    m, n = G.shape
    x = np.zeros(n)
    obj = C.c_double()
    st = Stats()
    o = make_opts(**kw)
    lib().orc_solve_miqp.restype = C.c_int
    s = lib().orc_solve_miqp(n, m, _p(P), _p(q), _p(G), _p(h), _p(lb), _p(ub), ib.ctypes.data_as(C.POINTER(C.c_ubyte)),
                             C.byref(o), _p(x), C.byref(obj), C.byref(st))
    return dict(status=STATUS[s], obj=obj.value, x=x if (np.isfinite(obj.value) or s == 4) else None, nodes=st.nodes,
                pivots=st.pivots, cuts=st.cuts, refactors=st.refactors, root_lp=st.root_lp,
                root_bound=st.root_bound, lower_bound=st.lower_bound)


def enumerate_milp(q, G, h, lb, ub, is_bin):
    q = np.ascontiguousarray(q, np.float64)
    G = np.ascontiguousarray(G, np.float64)
    h = np.ascontiguousarray(h, np.float64)
    lb = np.ascontiguousarray(lb, np.float64)
    ub = np.ascontiguousarray(ub, np.float64)
    ib = np.ascontiguousarray(is_bin, np.uint8)
    m, n = G.shape
    x = np.zeros(n)
    obj = C.c_double()
    s = lib().orc_enumerate_milp(n, m, _p(q), _p(G), _p(h), _p(lb), _p(ub),
                                 ib.ctypes.data_as(C.POINTER(C.c_ubyte)), _p(x), C.byref(obj))
    if s < 0:
        raise ValueError("too many binaries to enumerate")
    return dict(status=STATUS[s], obj=obj.value, x=x)


def condense(mats, dims, N):
    """C restatement of the condensing; returns the same dict layout as condense_np.condense"""
    d = Dims(dims["nx"], dims["nu"], dims["ndelta"], dims["nz"], dims["nmu"], dims["nomega"], dims["ny"],
             dims["nc"], dims.get("nu_l", 0), dims.get("nmu_l", 0))
    arrs = []
    for name in MAT_NAMES:
        a = mats.get(name)
        if a is None or np.size(a) == 0:
            arrs.append(None)
        else:
            arrs.append(np.ascontiguousarray(np.atleast_2d(np.asarray(a, np.float64))))
    nx, ny, nc, nw = dims["nx"], dims["ny"], dims["nc"], dims["nomega"]
    nv = dims["nu"] + dims["ndelta"] + dims["nz"] + dims["nmu"]
    shapes = dict(Phi_x=(N * nx, nx), Gamma_v=(N * nx, N * nv), Gamma_omega=(N * nx, N * nw), Gamma_5=(N * nx, 1),
                  L_x=(N * ny, nx), L_v=(N * ny, N * nv), L_omega=(N * ny, N * nw), L_5=(N * ny, 1),
                  H_x=(N * nc, nx), H_v=(N * nc, N * nv), H_omega=(N * nc, N * nw), H_5=(N * nc, 1))
    out = {k: np.zeros(s) for k, s in shapes.items()}
    order = ("Phi_x", "Gamma_v", "Gamma_omega", "Gamma_5", "L_x", "L_v", "L_omega", "L_5", "H_x", "H_v", "H_omega", "H_5")
    lib().orc_condense(C.byref(d), N, *[_p(a) for a in arrs], *[_p(out[k]) for k in order])
    return out
