"""ORACLE (test infrastructure, never shipped): probing-based big-M tightening of the per-step MLD rows

    E x + F1 u + F2 delta + F3 z + F4 omega + G y + Psi mu <= f5          (models/mld_model.py:459)

MLD models are produced by big-M translations of logic ([f<=0] <-> [delta=1], z = delta*f; the
example's grid model, modelling/micro_grid_models.py:141-170, is one).  With loose constants the LP
relaxation of the condensed problem is very weak.  This is the classical MIP presolve step
"probing + coefficient tightening" (Savelsbergh 1994, "Preprocessing and probing techniques for
mixed integer programming problems", sections 1.1, 3.2-3.3) applied to ONE step's rows with
(x, u, delta, z, omega, y, mu) all treated as variables: x, omega, y and continuous u/z are free,
mu >= 0, binaries in [0,1].  Every integer-feasible point of the original rows satisfies the
tightened rows and vice versa (only fractional points are cut off), so the MPC optimum is
unchanged; tests check that on solved instances.
"""
import numpy as np

_NAMES = ("E", "F1", "F2", "F3", "F4", "G", "Psi")


def _layout(d):
    """column blocks of one step's row matrix W = [E F1 F2 F3 F4 G Psi]"""
    sizes = [d["nx"], d["nu"], d["ndelta"], d["nz"], d["nomega"], d["ny"], d["nmu"]]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    return sizes, offs


def _propagate(W, c, lb, ub, is_int, passes=6):
    lb, ub = lb.copy(), ub.copy()
    m, n = W.shape
    for _ in range(passes):
        changed = False
        for i in range(m):
            js = np.nonzero(W[i])[0]
            if js.size == 0:
                continue
            a = W[i, js]
            lo_c = np.where(a > 0, a * lb[js], a * ub[js])
            ninf = np.isinf(lo_c)
            tot = lo_c[~ninf].sum()
            k = int(ninf.sum())
            if k == 0 and tot > c[i] + 1e-9 * max(1.0, abs(c[i])):
                return None
            for t, j in enumerate(js):
                if k == 0:
                    rest = tot - lo_c[t]
                elif k == 1 and ninf[t]:
                    rest = tot
                else:
                    continue
                b = (c[i] - rest) / a[t]
                if a[t] > 0:
                    if is_int[j]:
                        b = np.floor(b + 1e-9)
                    if b < ub[j] - 1e-12 * max(1.0, abs(b)):
                        ub[j] = b
                        changed = True
                else:
                    if is_int[j]:
                        b = np.ceil(b - 1e-9)
                    if b > lb[j] + 1e-12 * max(1.0, abs(b)):
                        lb[j] = b
                        changed = True
        if np.any(lb > ub + 1e-9):
            return None
        if not changed:
            break
    return lb, ub


def _max_activity(a, lb, ub, skip):
    """max of sum_{j != skip} a_j x_j over the box; +inf if unbounded"""
    hi = np.zeros_like(a)
    pos, neg = a > 0, a < 0
    hi[pos] = a[pos] * ub[pos]
    hi[neg] = a[neg] * lb[neg]
    hi[skip] = 0.0
    return float(hi.sum())


def tighten(mats, dims, nu_l=0, nmu_l=0, rounds=2):
    """returns a copy of `mats` with tightened F1/F2/Psi binary columns and f5 (other entries equal)"""
    d = dims
    sizes, offs = _layout(d)
    nc = d["nc"]
    ntot = int(offs[-1])
    if nc == 0 or ntot == 0:
        return dict(mats)

    def get(name, r, cdim):
        a = mats.get(name)
        if a is None or np.size(a) == 0:
            return np.zeros((r, cdim))
        return np.asarray(a, np.float64).reshape(r, cdim).copy()

    W = np.hstack([get(nm, nc, s) for nm, s in zip(_NAMES, sizes)])
    c = get("f5", nc, 1)[:, 0]
    lb = np.full(ntot, -np.inf)
    ub = np.full(ntot, np.inf)
    is_int = np.zeros(ntot, dtype=bool)
    ou, od, omu = int(offs[1]), int(offs[2]), int(offs[6])
    bin_cols = list(range(ou + d["nu"] - nu_l, ou + d["nu"])) + list(range(od, od + d["ndelta"])) + \
        list(range(omu + d["nmu"] - nmu_l, omu + d["nmu"]))
    lb[omu:omu + d["nmu"]] = 0.0                    # mu >= 0 (variables.py:221)
    for j in bin_cols:
        lb[j], ub[j], is_int[j] = 0.0, 1.0, True
    for _ in range(rounds):
        base = _propagate(W, c, lb, ub, is_int)
        if base is None:
            return dict(mats)                        # per-step rows infeasible on their own: leave alone
        lb, ub = base
        cond = {}
        for b in bin_cols:
            for v in (0.0, 1.0):
                l2, u2 = lb.copy(), ub.copy()
                if not (l2[b] <= v <= u2[b]):
                    cond[(b, v)] = None
                    continue
                l2[b] = u2[b] = v
                cond[(b, v)] = _propagate(W, c, l2, u2, is_int)
        for b in bin_cols:                           # a value whose probe is infeasible is excluded
            if cond[(b, 0.0)] is None and cond[(b, 1.0)] is not None:
                lb[b] = 1.0
            elif cond[(b, 1.0)] is None and cond[(b, 0.0)] is not None:
                ub[b] = 0.0
        for i in range(nc):
            for b in bin_cols:
                a = W[i, b]
                if a == 0.0:
                    continue
                if a > 0:                            # b = 0 side:  rest <= c_i
                    pr = cond[(b, 0.0)]
                    if pr is None:
                        continue
                    U = _max_activity(W[i], pr[0], pr[1], b)
                    if np.isfinite(U) and U < c[i] - 1e-12 * max(1.0, abs(c[i])):
                        dlt = c[i] - U
                        c[i] = U
                        W[i, b] = a - dlt
                else:                                # b = 1 side:  rest <= c_i - a
                    pr = cond[(b, 1.0)]
                    if pr is None:
                        continue
                    U = _max_activity(W[i], pr[0], pr[1], b)
                    if np.isfinite(U) and U < c[i] - a - 1e-12 * max(1.0, abs(c[i] - a)):
                        W[i, b] = c[i] - U
    out = dict(mats)
    for nm, s, o in zip(_NAMES, sizes, offs[:-1]):
        if s:
            out[nm] = W[:, int(o):int(o) + s].copy()
    out["f5"] = c.reshape(nc, 1).copy()
    return out
