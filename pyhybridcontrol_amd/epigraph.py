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
Rate atoms (``d<var>``) need lag states and are not covered (NotImplementedError).
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
            raise NotImplementedError("rate ('d<var>') atoms need lag states: not on the GPU path yet")
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
