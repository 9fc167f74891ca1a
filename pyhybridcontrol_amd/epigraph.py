"""L1 / Linf objective atoms on the GPU path by epigraph augmentation of the MLD model (host side, numpy only).

The reference hands ``cvx.norm1`` / ``cvx.norm_inf`` atoms to cvxpy (controllers/components/objective_atoms.py:334-363),
which introduces epigraph variables inside the solver interface.  Here the same reformulation is made on the *model*:
for an atom on variable ``var`` (one of x, u, delta, z, mu, y) with per-step coefficient matrix ``M``

    t >= M var ,  t >= -M var          (two blocks of rows in  E x + F1 u + F2 delta + F3 z + G y + Psi mu <= f5)

with new continuous auxiliaries ``t`` appended to ``z``; the atom becomes a *linear* atom on ``t``.  The augmented
system is an ordinary MLD model, so condensing, right-hand sides and the cut-and-branch kernel run unchanged.

Atom semantics follow the reference exactly, including its quirk that ``Linf`` with a vector or matrix weight is
evaluated as a per-step 1-norm (:355-363); the per-step max (:352-353) is only reached with no weight object at all,
which the string-keyed atom syntax (:453-496) cannot produce.
Rate atoms (``d<var>`` = var(k) - var(k-1)) become ordinary atoms on *rate outputs* of a model with lag states
(``augment_rates``); the value before the horizon, var(k-1), is the lag states' initial condition.
"""
import numpy as np

_CON = dict(x="E", u="F1", delta="F2", z="F3", mu="Psi", y="G")
_VDIM = dict(x="nx", u="nu", delta="ndelta", z="nz", mu="nmu", y="ny")


def _per_step_blocks(w, vd, N):
    """(vd, vd) block W0 if the (N vd, N vd) matrix weight is block diagonal with identical blocks, else None"""
    W0 = w[:vd, :vd]
    ref = np.kron(np.eye(N), W0)
    return W0 if np.allclose(w, ref, rtol=0, atol=1e-14 * max(1.0, np.abs(w).max())) else None


def plan(weights, dims, N):
    """epigraph blocks for the L1 / Linf atoms in ``weights`` ({(var, atype, wtype, rate): weight_N_tilde}).
    Returns a list of dicts: var, M (r, vd), S (r, nt) [t = S-expanded bound], cost (N, nt) per-step linear weights on t."""
    blocks = []
    for (var, atype, wtype, rate), w in weights.items():
        if atype not in ("L1", "Linf"):
            continue
        if rate:
            continue                           # rate atoms: rate_cost_and_blocks() below
        if var not in _CON:
            raise NotImplementedError("%s atom on '%s'" % (atype, var))
        vd = dims[_VDIM[var]]
        if vd == 0:
            continue
        if wtype == "vector":
            wk = np.abs(w[:, 0]).reshape(N, vd)
            # sum_k || w_k o var_k ||_1   (L1 :340-341; Linf with a weight is the same expression in the reference, :355-357)
            blocks.append(dict(var=var, M=np.eye(vd), S=np.eye(vd), cost=wk))
        else:
            W0 = _per_step_blocks(w, vd, N)
            if W0 is None:
                raise NotImplementedError("%s atom with a matrix weight that differs between horizon steps" % atype)
            blocks.append(dict(var=var, M=W0, S=np.eye(vd), cost=np.ones((N, vd))))   # sum_k || W0 var_k ||_1      (:343-344, 359-362)
    return blocks


def hard_block(dims, N):
    """rows  mu <= 0  (with mu >= 0: mu == 0) -- `disable_soft_constraints` of controller_base.py:466-471 as model rows"""
    k = dims["nmu"]
    return dict(var="mu", M=np.eye(k), S=np.zeros((k, 0)), cost=np.zeros((N, 0)), one_sided=True)


def augment(mats, dims, blocks):
    """(mats', dims', nt_total): the MLD system with the epigraph rows / auxiliaries of ``blocks`` appended"""
    nx, ny, nc, nz = dims["nx"], dims["ny"], dims["nc"], dims["nz"]
    nt = sum(b["S"].shape[1] for b in blocks)
    nr = sum((1 if b.get("one_sided") else 2) * b["M"].shape[0] for b in blocks)

    def get(name, r, c):
        a = mats.get(name)
        if a is None or np.size(a) == 0:
            return np.zeros((r, c))
        return np.asarray(a, dtype=np.float64).reshape(r, c)

    out = {k: mats.get(k) for k in ("A", "B1", "B2", "B4", "b5", "C", "D1", "D2", "D4", "d5")}
    out["B3"] = np.hstack([get("B3", nx, nz), np.zeros((nx, nt))])
    out["D3"] = np.hstack([get("D3", ny, nz), np.zeros((ny, nt))])
    con = {k: get(k, nc, dims[v]) for k, v in (("E", "nx"), ("F1", "nu"), ("F2", "ndelta"), ("F4", "nomega"), ("G", "ny"), ("Psi", "nmu"))}
    con["F3"] = np.hstack([get("F3", nc, nz), np.zeros((nc, nt))])
    f5 = get("f5", nc, 1)
    new = {k: np.zeros((nr, a.shape[1])) for k, a in con.items()}
    r0, t0 = 0, nz
    for b in blocks:
        M, S = b["M"], b["S"]
        r, k = M.shape[0], S.shape[1]
        name = _CON[b["var"]]
        col0 = 0
        new[name][r0:r0 + r, col0:col0 + M.shape[1]] += M           #  M var - S t <= 0
        new["F3"][r0:r0 + r, t0:t0 + k] -= S
        r0 += r
        if not b.get("one_sided"):
            new[name][r0:r0 + r, col0:col0 + M.shape[1]] -= M       # -M var - S t <= 0
            new["F3"][r0:r0 + r, t0:t0 + k] -= S
            r0 += r
        t0 += k
    for k in con:
        out[k] = np.vstack([con[k], new[k]])
    out["f5"] = np.vstack([f5, np.zeros((nr, 1))])
    d2 = dict(dims)
    d2["nz"] = nz + nt
    d2["nc"] = nc + nr
    return out, d2, nt


def lift_cost(cost, dims, dims2, N, blocks):
    """cost dict over the original v layout -> the augmented layout, plus the linear weights on the auxiliaries"""
    nv = dims["nu"] + dims["ndelta"] + dims["nz"] + dims["nmu"]
    nv2 = dims2["nu"] + dims2["ndelta"] + dims2["nz"] + dims2["nmu"]
    nt = nv2 - nv
    head = dims["nu"] + dims["ndelta"] + dims["nz"]
    pos = np.concatenate([np.arange(head), np.arange(head, nv) + nt])
    vmap = np.concatenate([k * nv2 + pos for k in range(N)])          # original v_tilde index -> augmented index
    out = dict(cost)
    lin = np.zeros(N * nv2)
    if cost.get("lin_v") is not None:
        lin[vmap] = np.asarray(cost["lin_v"], dtype=np.float64).ravel()
    t0 = head
    for b in blocks:
        k = b["S"].shape[1]
        for s in range(N):
            lin[s * nv2 + t0:s * nv2 + t0 + k] += b["cost"][s]
        t0 += k
    out["lin_v"] = lin
    if cost.get("quad_v") is not None:
        Q = np.zeros((N * nv2, N * nv2))
        Q[np.ix_(vmap, vmap)] = cost["quad_v"]
        out["quad_v"] = Q
    return out, vmap


# ---- rate atoms ('d<var>': var(k) - var(k-1), objective_atoms.py:296-304) through lag states ------------------------
_RATE_IN = dict(u=("B1", "D1", "nu"), delta=("B2", "D2", "ndelta"), z=("B3", "D3", "nz"))


def rate_vars(weights):
    """variables that carry a rate atom, in a fixed order"""
    order = ("x", "u", "delta", "z", "y")
    found = {var for (var, _a, _w, rate) in weights if rate}
    bad = found - set(order)
    if bad:
        raise NotImplementedError("rate atom on %s (mu / v do not enter the state equation)" % sorted(bad))
    return [v for v in order if v in found]


def augment_rates(mats, dims, rvars):
    """(mats', dims', info): lag states  x_lag(k+1) = var(k)  and rate outputs  y_d(k) = var(k) - x_lag(k)  appended to the
    MLD system for every variable in ``rvars``.  info[var] = (lag state offset, rate output offset, dim)."""
    nx, ny, nc = dims["nx"], dims["ny"], dims["nc"]
    cols = dict(A="nx", B1="nu", B2="ndelta", B3="nz", B4="nomega", C="nx", D1="nu", D2="ndelta", D3="nz", D4="nomega")

    def get(name, r, c):
        a = mats.get(name)
        if a is None or np.size(a) == 0:
            return np.zeros((r, c))
        return np.asarray(a, dtype=np.float64).reshape(r, c)

    S = {k: get(k, nx, dims[v]) for k, v in cols.items() if k[0] in "AB"}
    O = {k: get(k, ny, dims[v]) for k, v in cols.items() if k[0] in "CD"}
    b5, d5 = get("b5", nx, 1), get("d5", ny, 1)
    dim_of = dict(x=nx, y=ny, u=dims["nu"], delta=dims["ndelta"], z=dims["nz"])
    nl = sum(dim_of[v] for v in rvars)
    info, lo = {}, 0
    s_rows = {k: [a] for k, a in S.items()}
    o_rows = {k: [a] for k, a in O.items()}
    b_rows, d_rows = [b5], [d5]
    lagA, lagC = [], []          # (rows x nl) blocks acting on the lag states
    for v in rvars:
        k = dim_of[v]
        info[v] = (nx + lo, ny + lo, k)
        new_s = {kk: np.zeros((k, a.shape[1])) for kk, a in S.items()}
        new_o = {kk: np.zeros((k, a.shape[1])) for kk, a in O.items()}
        nb5, nd5 = np.zeros((k, 1)), np.zeros((k, 1))
        if v == "x":
            new_s["A"][:, :] = np.eye(nx); new_o["C"][:, :] = np.eye(nx)
        elif v == "y":
            for src, dst in (("C", "A"), ("D1", "B1"), ("D2", "B2"), ("D3", "B3"), ("D4", "B4")):
                new_s[dst][:, :] = O[src]
            nb5[:] = d5
            for src in ("C", "D1", "D2", "D3", "D4"):
                new_o[src][:, :] = O[src]
            nd5[:] = d5
        else:
            bname, dname, _ = _RATE_IN[v]
            new_s[bname][:, :] = np.eye(k); new_o[dname][:, :] = np.eye(k)
        for kk in S:
            s_rows[kk].append(new_s[kk])
        for kk in O:
            o_rows[kk].append(new_o[kk])
        b_rows.append(nb5); d_rows.append(nd5)
        L = np.zeros((k, nl)); L[:, lo:lo + k] = -np.eye(k)
        lagC.append(L)
        lo += k
    out = {}
    for kk in S:
        out[kk] = np.vstack(s_rows[kk])
    for kk in O:
        out[kk] = np.vstack(o_rows[kk])
    out["A"] = np.hstack([out["A"], np.zeros((nx + nl, nl))])               # lag states feed nothing but the rate outputs
    out["C"] = np.hstack([out["C"], np.vstack([np.zeros((ny, nl))] + lagC)])
    out["b5"], out["d5"] = np.vstack(b_rows), np.vstack(d_rows)
    out["E"] = np.hstack([get("E", nc, nx), np.zeros((nc, nl))])
    out["G"] = np.hstack([get("G", nc, ny), np.zeros((nc, nl))])
    for kk in ("F1", "F2", "F3", "F4", "f5", "Psi"):
        out[kk] = mats.get(kk)
    d2 = dict(dims)
    d2["nx"], d2["ny"] = nx + nl, ny + nl
    return out, d2, info


def lift_xy_cost(cost, dims, dims2, N):
    """cost dict over the original x / y layouts -> the layouts with lag states / rate outputs appended per step"""
    out = dict(cost)
    for key, d0, d1 in (("x", dims["nx"], dims2["nx"]), ("y", dims["ny"], dims2["ny"])):
        idx = np.concatenate([k * d1 + np.arange(d0) for k in range(N)]) if d0 else np.zeros(0, dtype=int)
        lin = np.zeros(N * d1)
        if cost.get("lin_" + key) is not None and d0:
            lin[idx] = np.asarray(cost["lin_" + key], dtype=np.float64).ravel()
        out["lin_" + key] = lin
        if cost.get("quad_" + key) is not None:
            Q = np.zeros((N * d1, N * d1))
            Q[np.ix_(idx, idx)] = cost["quad_" + key]
            out["quad_" + key] = Q
    return out


def rate_cost_and_blocks(weights, dims2, info, N, cost):
    """adds the rate atoms to ``cost`` (already in the dims2 layouts) as atoms on the rate outputs; returns the epigraph
    blocks their L1 / Linf members need"""
    ny2 = dims2["ny"]
    blocks = []
    for (var, atype, wtype, rate), w in weights.items():
        if not rate:
            continue
        _, yo, k = info[var]
        rows = np.concatenate([s * ny2 + yo + np.arange(k) for s in range(N)])
        if atype == "Linear":
            cost["lin_y"][rows] += w[:, 0] if wtype == "vector" else w.sum(axis=0)
        elif atype in ("Quadratic", "L22"):
            W = np.diag(w[:, 0] ** 2) if wtype == "vector" else w
            if cost.get("quad_y") is None:
                cost["quad_y"] = np.zeros((N * ny2, N * ny2))
            cost["quad_y"][np.ix_(rows, rows)] += W
        else:                                  # L1 / Linf (= per-step 1-norm, see plan())
            sel = np.zeros((k, ny2)); sel[:, yo:yo + k] = np.eye(k)
            if wtype == "vector":
                blocks.append(dict(var="y", M=sel, S=np.eye(k), cost=np.abs(w[:, 0]).reshape(N, k)))
            else:
                W0 = _per_step_blocks(w, k, N)
                if W0 is None:
                    raise NotImplementedError("%s rate atom with a matrix weight that differs between horizon steps" % atype)
                blocks.append(dict(var="y", M=W0 @ sel, S=np.eye(k), cost=np.ones((N, k))))
    return blocks
