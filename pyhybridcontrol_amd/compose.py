"""Fusing device MLD models into a grid MLD model (host side, numpy only).

The reference's example composes a micro-grid at the cvxpy level: every device controller contributes its constraints
and objective to the grid controller's problem and the grid's disturbance input is *constrained to equal* the stacked
device powers (examples/residential_mg_with_pv_and_dewhs/modelling/micro_grid_agents.py:625-709, grid model
micro_grid_models.py:137-172).  Numerically that is one MLD system: block-diagonal device dynamics, and the grid's rows
with its disturbance columns substituted by the devices' output equations.  ``fuse`` builds that system, so that a whole
grid is ONE instance for the GPU path (condense -> rhs -> cut-and-branch) instead of a bag of cvxpy expressions.
"""
import numpy as np

_VARS = (("x", "nx"), ("u", "nu"), ("delta", "ndelta"), ("z", "nz"), ("mu", "nmu"), ("omega", "nomega"), ("y", "ny"))
_S = dict(x="A", u="B1", delta="B2", z="B3", omega="B4")
_O = dict(x="C", u="D1", delta="D2", z="D3", omega="D4")
_K = dict(x="E", u="F1", delta="F2", z="F3", omega="F4", y="G", mu="Psi")


def _get(mats, name, r, c):
    a = mats.get(name)
    if a is None or np.size(a) == 0:
        return np.zeros((r, c))
    return np.asarray(a, dtype=np.float64).reshape(r, c)


def fuse(devices, grid):
    """devices: list of (mats, dims); grid: (mats, dims) whose first sum(ny_i) disturbance entries are the device
    outputs (in device order; remaining entries stay an external disturbance).  Returns (mats, dims, layout) of the
    fused MLD system with variables [dev_1 .. dev_n, grid] in every block; layout[name] = list of (offset, size) per
    sub-system (grid last) for x, u, delta, z, mu, omega, y and the constraint rows ('c')."""
    subs = list(devices) + [grid]
    n_dev = len(devices)
    gm, gd = grid
    ny_dev = sum(d["ny"] for _, d in devices)
    if ny_dev > gd["nomega"]:
        raise ValueError("the grid model has %d disturbance inputs, the devices deliver %d outputs" % (gd["nomega"], ny_dev))
    for _, d in subs:
        if d.get("nu_l", 0) not in (0, d["nu"]) or d.get("nmu_l", 0) not in (0, d["nmu"]):
            raise NotImplementedError("sub-systems with mixed continuous / binary inputs (binaries must stay trailing)")
    cont_first = sorted(range(len(subs)), key=lambda i: (subs[i][1].get("nu_l", 0) > 0, i))    # continuous u blocks first
    order = dict(u=cont_first, mu=sorted(range(len(subs)), key=lambda i: (subs[i][1].get("nmu_l", 0) > 0, i)))
    layout, tot = {}, {}
    for v, key in _VARS:
        sizes = [d[key] for _, d in subs]
        if v == "omega":
            sizes[-1] = gd["nomega"] - ny_dev                    # the grid keeps only its external disturbance entries
        seq = order.get(v, range(len(subs)))
        offs, o = [None] * len(subs), 0
        for i in seq:
            offs[i] = (o, sizes[i]); o += sizes[i]
        layout[v], tot[v] = offs, o
    rows_c, o = [], 0
    for _, d in subs:
        rows_c.append((o, d["nc"])); o += d["nc"]
    layout["c"], tot["c"] = rows_c, o

    def zeros(rk, ck):
        return np.zeros((tot[rk], tot[ck]))
    S = {v: zeros("x", v) for v in _S}
    O = {v: zeros("y", v) for v in _O}
    K = {v: zeros("c", v) for v in _K}
    b5, d5, f5 = np.zeros((tot["x"], 1)), np.zeros((tot["y"], 1)), np.zeros((tot["c"], 1))

    def put(M, rk, i, ck, j, block):
        (r0, rn), (c0, cn) = layout[rk][i], layout[ck][j]
        M[r0:r0 + rn, c0:c0 + cn] += block

    # device blocks (and the grid's own block, its external disturbance columns only)
    for i, (m, d) in enumerate(subs):
        own_w = slice(ny_dev, gd["nomega"]) if i == n_dev else slice(0, d["nomega"])
        for v in _S:
            blk = _get(m, _S[v], d["nx"], d["n" + v])
            put(S[v], "x", i, v, i, blk[:, own_w] if v == "omega" else blk)
            blk = _get(m, _O[v], d["ny"], d["n" + v])
            put(O[v], "y", i, v, i, blk[:, own_w] if v == "omega" else blk)
        for v in _K:
            blk = _get(m, _K[v], d["nc"], d["n" + v])
            put(K[v], "c", i, v, i, blk[:, own_w] if v == "omega" else blk)
        (r0, rn) = layout["x"][i]; b5[r0:r0 + rn] += _get(m, "b5", d["nx"], 1)
        (r0, rn) = layout["y"][i]; d5[r0:r0 + rn] += _get(m, "d5", d["ny"], 1)
        (r0, rn) = layout["c"][i]; f5[r0:r0 + rn] += _get(m, "f5", d["nc"], 1)
    # the grid's disturbance columns that carry device outputs: substitute  y_dev = C x + D1 u + D2 delta + D3 z + D4 omega + d5
    g = n_dev
    col = 0
    for i, (m, d) in enumerate(devices):
        sl = slice(col, col + d["ny"]); col += d["ny"]
        for fam, RK, const, wname, wrows in ((S, "x", b5, "B4", gd["nx"]), (O, "y", d5, "D4", gd["ny"]), (K, "c", f5, "F4", gd["nc"])):
            W = _get(gm, wname, wrows, gd["nomega"])[:, sl]
            for v in _O:
                put(fam[v], RK, g, v, i, W @ _get(m, _O[v], d["ny"], d["n" + v]))
            (r0, rn) = layout[RK][g]
            contrib = W @ _get(m, "d5", d["ny"], 1)
            const[r0:r0 + rn] += -contrib if fam is K else contrib      # constraints:  ... + W d5 <= f5  ->  f5 - W d5
    mats = dict(A=S["x"], B1=S["u"], B2=S["delta"], B3=S["z"], B4=S["omega"], b5=b5,
                C=O["x"], D1=O["u"], D2=O["delta"], D3=O["z"], D4=O["omega"], d5=d5,
                E=K["x"], F1=K["u"], F2=K["delta"], F3=K["z"], F4=K["omega"], f5=f5, G=K["y"], Psi=K["mu"])
    dims = dict(nx=tot["x"], nu=tot["u"], ndelta=tot["delta"], nz=tot["z"], nmu=tot["mu"], nomega=tot["omega"], ny=tot["y"], nc=tot["c"],
                nu_l=sum(d.get("nu_l", 0) for _, d in subs), nmu_l=sum(d.get("nmu_l", 0) for _, d in subs))
    return mats, dims, layout
