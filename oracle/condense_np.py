"""ORACLE (test infrastructure, never shipped, never on the product path).

numpy restatement of the reference's condensing / variable-layout / cost-assembly arithmetic for the
MPC hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned against golden vectors produced by the reference itself (oracle/gen_golden.py ->
tests/golden/*.npz); see tests/test_oracle_golden.py.

Each function cites the reference file:line it follows (paths relative to /root/reference).
Written as plain loops over block indices (no stride tricks) so that it is an independent statement
of *what* is computed, not a copy of *how* the reference computes it.
"""
import re

import numpy as np

STATE_INPUT_MATS = ("A", "B1", "B2", "B3", "B4", "b5")
OUTPUT_MATS = ("C", "D1", "D2", "D3", "D4", "d5")
CONS_MATS = ("E", "F1", "F2", "F3", "F4", "f5", "G", "Psi")


def mld_dims(mats):
    """Dimensions from matrix shapes -- models/mld_model.py:149-168 (_mld_dim_map)."""

    def rows(names):
        return max([np.atleast_2d(mats[n]).shape[0] for n in names if n in mats and np.size(mats[n])] + [0])

    def cols(names):
        return max([np.atleast_2d(mats[n]).shape[1] for n in names if n in mats and np.size(mats[n])] + [0])

    d = dict(nx=rows(STATE_INPUT_MATS), ny=rows(OUTPUT_MATS), nc=rows(CONS_MATS),
             nu=cols(("B1", "D1", "F1")), ndelta=cols(("B2", "D2", "F2")), nz=cols(("B3", "D3", "F3")),
             nomega=cols(("B4", "D4", "F4")), nmu=cols(("Psi",)))
    d["nv"] = d["nu"] + d["ndelta"] + d["nz"] + d["nmu"]
    return d


def pad_mats(mats, d):
    """Missing matrices -> zeros of (sys_dim, var_dim) -- models/mld_model.py:910-928; C defaults to
    I(nx) when absent (models/mld_model.py:515-520)."""
    shp = dict(A=("nx", "nx"), B1=("nx", "nu"), B2=("nx", "ndelta"), B3=("nx", "nz"), B4=("nx", "nomega"),
               b5=("nx", 1), C=("ny", "nx"), D1=("ny", "nu"), D2=("ny", "ndelta"), D3=("ny", "nz"),
               D4=("ny", "nomega"), d5=("ny", 1), E=("nc", "nx"), F1=("nc", "nu"), F2=("nc", "ndelta"),
               F3=("nc", "nz"), F4=("nc", "nomega"), f5=("nc", 1), G=("nc", "ny"), Psi=("nc", "nmu"))
    out = {}
    for name, (r, c) in shp.items():
        rr = d[r] if isinstance(r, str) else r
        cc = d[c] if isinstance(c, str) else c
        m = mats.get(name)
        if m is None or np.size(m) == 0:
            out[name] = np.zeros((rr, cc))
        else:
            m = np.asarray(m, dtype=np.float64)
            if m.ndim == 0:
                m = m.reshape(1, 1)
            elif m.ndim == 1:
                m = m.reshape(-1, 1)
            assert m.shape == (rr, cc), (name, m.shape, (rr, cc))
            out[name] = m
    return out


def var_types(d, nu_l=0, nmu_l=0):
    """Per-step binary mask of v=[u;delta;z;mu] -- models/mld_model.py:294-345: binaries are the
    trailing nu_l (nmu_l) entries of u (mu); every delta is binary; z never is."""
    mask = np.zeros(d["nv"], dtype=bool)
    o = 0
    mask[o + d["nu"] - nu_l:o + d["nu"]] = True
    o += d["nu"]
    mask[o:o + d["ndelta"]] = True
    o += d["ndelta"] + d["nz"]
    mask[o + d["nmu"] - nmu_l:o + d["nmu"]] = True
    return mask


def block_toeplitz_lower(col_blocks):
    """Lower block-Toeplitz with zero upper part: block (i,j) = col_blocks[i-j] for i>=j.
    utils/matrix_utils.py:117-161 with r_tup = zeros (as called from
    controllers/components/mld_evolution_matrices.py:495-499)."""
    N = len(col_blocks)
    r, c = col_blocks[0].shape
    out = np.zeros((N * r, N * c))
    for i in range(N):
        for j in range(i + 1):
            out[i * r:(i + 1) * r, j * c:(j + 1) * c] = col_blocks[i - j]
    return out


def block_diag_rep(block, N):
    """N copies of `block` on the diagonal -- utils/matrix_utils.py:55-81."""
    r, c = block.shape
    out = np.zeros((N * r, N * c))
    for i in range(N):
        out[i * r:(i + 1) * r, i * c:(i + 1) * c] = block
    return out


def condense(mats, N_tilde):
    """All twelve *_N_tilde evolution matrices.
    controllers/components/mld_evolution_matrices.py:137-244, 253-332, 467-527."""
    d = mld_dims(mats)
    m = pad_mats(mats, d)
    nx, nv, nw, ny, nc = d["nx"], d["nv"], d["nomega"], d["ny"], d["nc"]
    N = N_tilde
    A = m["A"]
    Bv = np.hstack([m["B1"], m["B2"], m["B3"], np.zeros((nx, d["nmu"]))])  # :291 '_zeros_Psi_state_input'
    Dv = np.hstack([m["D1"], m["D2"], m["D3"], np.zeros((ny, d["nmu"]))])  # :355
    Fv = np.hstack([m["F1"], m["F2"], m["F3"], m["Psi"]])                  # :411
    # A^0 .. A^{N-1}  (:253-272)
    Apow = [np.eye(nx)]
    for _ in range(N - 1):
        Apow.append(Apow[-1] @ A)
    Phi_x = np.vstack(Apow) if nx else np.zeros((0, nx))
    # col = [0, A^0 B, A^1 B, ...]  (:495-497)
    col_v = [np.zeros((nx, nv))] + [Apow[k] @ Bv for k in range(N - 1)]
    col_w = [np.zeros((nx, nw))] + [Apow[k] @ m["B4"] for k in range(N - 1)]
    col_5 = [np.zeros((nx, 1))] + [Apow[k] @ m["b5"] for k in range(N - 1)]
    Gamma_v = block_toeplitz_lower(col_v)
    Gamma_w = block_toeplitz_lower(col_w)
    Gamma_5 = block_toeplitz_lower(col_5) @ np.ones((N, 1))                # :331-332
    Ct, Dvt, D4t = block_diag_rep(m["C"], N), block_diag_rep(Dv, N), block_diag_rep(m["D4"], N)
    d5t = np.tile(m["d5"], (N, 1))
    L_x = Ct @ Phi_x                                                        # :186-189
    L_v = Ct @ Gamma_v + Dvt
    L_w = Ct @ Gamma_w + D4t
    L_5 = Ct @ Gamma_5 + d5t
    Et, Fvt, F4t, Gt = (block_diag_rep(m["E"], N), block_diag_rep(Fv, N), block_diag_rep(m["F4"], N),
                        block_diag_rep(m["G"], N))
    f5t = np.tile(m["f5"], (N, 1))
    H_x = -(Et @ Phi_x + Gt @ L_x)                                          # :237-240
    H_v = Et @ Gamma_v + Fvt + Gt @ L_v
    H_w = -(Et @ Gamma_w + F4t + Gt @ L_w)
    H_5 = f5t - (Et @ Gamma_5 + Gt @ L_5)
    return dict(Phi_x=Phi_x, Gamma_v=Gamma_v, Gamma_omega=Gamma_w, Gamma_5=Gamma_5,
                L_x=L_x, L_v=L_v, L_omega=L_w, L_5=L_5, H_x=H_x, H_v=H_v, H_omega=H_w, H_5=H_5, dims=d)


def condense_tv(mats_list, N_tilde=None):
    """Evolution matrices of a time-VARYING horizon: mats_list[k] is the numeric MLD model of horizon step k
    (`mld_numeric_tilde`, models/mld_model.py:1210-1227; time-varying branch of `_gen_A_pow_tilde`,
    controllers/components/mld_evolution_matrices.py:265-272).

    The reference's own branch is unreachable (typo at controllers/controller_utils.py:26) and accumulates its
    products in an order that is only right for a time-invariant model, so there is no reference behaviour to pin;
    what is restated here is the intent -- the linear time-varying recursion
        x(k+1) = A_k x(k) + [B1 B2 B3 0]_k v(k) + B4_k w(k) + b5_k ,   y(k) = C_k x(k) + D_k v(k) + ... ,
        E_k x(k) + F_k v(k) + F4_k w(k) + G_k y(k) <= f5_k
    written block by block: Phi_x[i] = A_{i-1} ... A_0, Gamma[i][j] = A_{i-1} ... A_{j+1} B_j (j < i), same sign
    conventions as condense().  With identical step models it reproduces condense() exactly."""
    N = len(mats_list) if N_tilde is None else N_tilde
    d = mld_dims(mats_list[0])
    ms = [pad_mats(m, d) for m in mats_list[:N]]
    nx, nv, nw, ny, nc = d["nx"], d["nv"], d["nomega"], d["ny"], d["nc"]
    Bv = [np.hstack([m["B1"], m["B2"], m["B3"], np.zeros((nx, d["nmu"]))]) for m in ms]
    Dv = [np.hstack([m["D1"], m["D2"], m["D3"], np.zeros((ny, d["nmu"]))]) for m in ms]
    Fv = [np.hstack([m["F1"], m["F2"], m["F3"], m["Psi"]]) for m in ms]
    Phi_x = np.zeros((N * nx, nx)); Gv = np.zeros((N * nx, N * nv)); Gw = np.zeros((N * nx, N * nw)); G5 = np.zeros((N * nx, 1))
    P = np.eye(nx); s5 = np.zeros((nx, 1))
    for i in range(N):
        Phi_x[i * nx:(i + 1) * nx] = P
        G5[i * nx:(i + 1) * nx] = s5
        s5 = ms[i]["A"] @ s5 + ms[i]["b5"]
        P = ms[i]["A"] @ P
    for j in range(N):
        Pv, Pw = Bv[j], ms[j]["B4"]
        for i in range(j + 1, N):
            Gv[i * nx:(i + 1) * nx, j * nv:(j + 1) * nv] = Pv
            Gw[i * nx:(i + 1) * nx, j * nw:(j + 1) * nw] = Pw
            Pv, Pw = ms[i]["A"] @ Pv, ms[i]["A"] @ Pw
    L_x = np.zeros((N * ny, nx)); L_v = np.zeros((N * ny, N * nv)); L_w = np.zeros((N * ny, N * nw)); L_5 = np.zeros((N * ny, 1))
    H_x = np.zeros((N * nc, nx)); H_v = np.zeros((N * nc, N * nv)); H_w = np.zeros((N * nc, N * nw)); H_5 = np.zeros((N * nc, 1))
    for i in range(N):
        m = ms[i]
        rx, ry, rc = slice(i * nx, (i + 1) * nx), slice(i * ny, (i + 1) * ny), slice(i * nc, (i + 1) * nc)
        L_x[ry] = m["C"] @ Phi_x[rx]
        L_v[ry] = m["C"] @ Gv[rx]; L_v[ry, i * nv:(i + 1) * nv] += Dv[i]
        L_w[ry] = m["C"] @ Gw[rx]; L_w[ry, i * nw:(i + 1) * nw] += m["D4"]
        L_5[ry] = m["C"] @ G5[rx] + m["d5"]
        H_x[rc] = -(m["E"] @ Phi_x[rx] + m["G"] @ L_x[ry])
        H_v[rc] = m["E"] @ Gv[rx] + m["G"] @ L_v[ry]; H_v[rc, i * nv:(i + 1) * nv] += Fv[i]
        Hw = m["E"] @ Gw[rx] + m["G"] @ L_w[ry]; Hw[:, i * nw:(i + 1) * nw] += m["F4"]
        H_w[rc] = -Hw
        H_5[rc] = m["f5"] - (m["E"] @ G5[rx] + m["G"] @ L_5[ry])
    return dict(Phi_x=Phi_x, Gamma_v=Gv, Gamma_omega=Gw, Gamma_5=G5, L_x=L_x, L_v=L_v, L_omega=L_w, L_5=L_5,
                H_x=H_x, H_v=H_v, H_omega=H_w, H_5=H_5, dims=d)


# ----------------------------------------------------------------------------- objective atoms
_ATOM_PAT = re.compile(r"(Linear)|(Quadratic)|([L](1|(22)|(inf)))")
_RATE_PAT = re.compile(r"[dD][^e]")
VAR_NAMES = ("x", "u", "delta", "z", "omega", "y", "mu", "v")


def parse_atom_key(key):
    """'<q|Q>[_Linear|_Quadratic|..]_[d]<var>[_N_tilde|_N_p|_f]' -> (weight_type, atom_type, var,
    is_rate, post_fix) -- controllers/components/objective_atoms.py:453-471."""
    info = key.split("_")
    wtype = "vector" if "".join(info[0:1]).islower() else "matrix"
    atype = "".join(info[1:2]).capitalize()
    if not _ATOM_PAT.search(atype):
        atype = "Linear" if wtype == "vector" else "Quadratic"
        var = "".join(info[1:2]).lower()
        post = "_".join(info[2:])
    else:
        var = "".join(info[2:3]).lower()
        post = "_".join(info[3:])
    rate = False
    if _RATE_PAT.search(var):
        var = var[1:]
        rate = True
    if var not in VAR_NAMES or post not in ("N_p", "N_tilde", "f", ""):
        raise ValueError("weight_name: %r is not valid" % key)
    return wtype, atype, var, rate, post


def _col(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 0:
        return a.reshape(1, 1)
    if a.ndim == 1:
        return a.reshape(-1, 1)
    return a


def tile_vector_weight(value, var_dim, length):
    """objective_atoms.py:118-137."""
    value = _col(value)
    if value.shape[1] != 1:
        raise ValueError("column dim of vector weight must be 1")
    if value.shape[0] == var_dim * length:
        return value.copy()
    if value.shape[0] == var_dim:
        return np.tile(value, (length, 1))
    raise ValueError("bad row dim for vector weight")


def tile_matrix_weight(value, var_dim, length):
    """objective_atoms.py:185-206."""
    value = _col(value)
    if value.shape[0] != value.shape[1]:
        raise ValueError("matrix weight must be square")
    if value.shape[0] == var_dim * length:
        return value.copy()
    if value.shape[0] == var_dim:
        return block_diag_rep(value, length)
    raise ValueError("bad dim for matrix weight")


def build_weights(atoms, dims, N_p, N_tilde):
    """Parsed + tiled weights: dict (var, atom_type, weight_type, rate) -> weight_N_tilde.
    objective_atoms.py:421-521 (update/_set_atom) with 76-206 (weight classes): a key without
    post-fix addresses N_tilde if its row count is var_dim or var_dim*N_tilde, else N_p (:480-485);
    '_f' overwrites the last step (:102-105,169-172); all-zero weights are dropped (:508,519-520)."""
    out = {}
    dim_of = dict(x=dims["nx"], u=dims["nu"], delta=dims["ndelta"], z=dims["nz"], omega=dims["nomega"],
                  y=dims["ny"], mu=dims["nmu"], v=dims["nv"])
    for key, value in atoms.items():
        if value is None:
            continue
        wtype, atype, var, rate, post = parse_atom_key(key)
        value = _col(value)
        vd = dim_of[var]
        if post:
            which = post
        elif value.shape[0] == vd or value.shape[0] == vd * N_tilde:
            which = "N_tilde"
        else:
            which = "N_p"
        ident = (var, atype, wtype, rate)
        w = out.get(ident)
        if w is None:
            if np.all(np.isclose(value, 0.0)):
                continue
            w = np.zeros((N_tilde * vd, 1)) if wtype == "vector" else np.zeros((N_tilde * vd, N_tilde * vd))
        if wtype == "vector":
            if which == "N_tilde":
                w[:] = tile_vector_weight(value, vd, N_tilde)
            elif which == "N_p":
                w[:N_p * vd, :1] = tile_vector_weight(value, vd, N_p)
            else:
                if value.shape != (vd, 1):
                    raise ValueError("terminal weight dim")
                w[-vd:, :1] = value
        else:
            if which == "N_tilde":
                w[:] = tile_matrix_weight(value, vd, N_tilde)
            elif which == "N_p":
                w[:N_p * vd, :N_p * vd] = tile_matrix_weight(value, vd, N_p)
            else:
                if value.shape != (vd, vd):
                    raise ValueError("terminal weight dim")
                w[-vd:, -vd:] = value
        if np.all(np.isclose(w, 0.0)):
            out.pop(ident, None)
        else:
            out[ident] = w
    return out


def _selector(dims, N_tilde, var):
    """Rows of v_tilde = [u0;d0;z0;mu0;u1;...] (variables.py:226-241, column-major reshape) that
    make up `var`_tilde (step-major stacking of that variable)."""
    nv = dims["nv"]
    offs = dict(u=(0, dims["nu"]), delta=(dims["nu"], dims["ndelta"]),
                z=(dims["nu"] + dims["ndelta"], dims["nz"]),
                mu=(dims["nu"] + dims["ndelta"] + dims["nz"], dims["nmu"]), v=(0, nv))
    o, n = offs[var]
    idx = np.concatenate([np.arange(k * nv + o, k * nv + o + n) for k in range(N_tilde)]) if n else np.zeros(0, int)
    return idx.astype(int)


def assemble_cost(weights, evo, dims, N_tilde):
    """Standard-form cost  1/2 v'Pv + (q0 + Qx x_k + Qw omega)'v + r(x_k,omega).
    Returns P (n,n), q0 (n,), Qx (n,nx), Qw (n,N nw) and the pieces of the constant term.
    Linear atom: w' var (vector) / sum(W var) (matrix); Quadratic: ||w o var||^2 (vector) /
    var' W var (matrix) -- objective_atoms.py:308-331; x,y are the affine maps of
    variables.py:259-275.  Only Linear/Quadratic non-rate atoms."""
    n = N_tilde * dims["nv"]
    nx, nw = dims["nx"], dims["nomega"]
    P = np.zeros((n, n))
    q0 = np.zeros(n)
    Qx = np.zeros((n, nx))
    Qw = np.zeros((n, N_tilde * nw))
    # constant term r = r0 + rx'x + rw'w + 1/2 [x;w]' R [x;w]  (kept as pieces; tests evaluate by value)
    const_terms = []
    for (var, atype, wtype, rate), w in weights.items():
        if rate or atype not in ("Linear", "Quadratic"):
            raise NotImplementedError("only Linear/Quadratic non-rate atoms on the hot path")
        # affine map var_tilde = M v + Mx x + Mw w + m0
        if var in ("u", "delta", "z", "mu", "v"):
            idx = _selector(dims, N_tilde, var)
            M = np.zeros((idx.size, n))
            M[np.arange(idx.size), idx] = 1.0
            Mx = np.zeros((idx.size, nx))
            Mw = np.zeros((idx.size, N_tilde * nw))
            m0 = np.zeros((idx.size, 1))
        elif var == "x":
            M, Mx, Mw, m0 = evo["Gamma_v"], evo["Phi_x"], evo["Gamma_omega"], evo["Gamma_5"]
        elif var == "y":
            M, Mx, Mw, m0 = evo["L_v"], evo["L_x"], evo["L_omega"], evo["L_5"]
        elif var == "omega":
            k = N_tilde * nw
            M, Mx, Mw, m0 = np.zeros((k, n)), np.zeros((k, nx)), np.eye(k), np.zeros((k, 1))
        else:
            raise ValueError(var)
        if atype == "Linear":
            lw = w[:, 0] if wtype == "vector" else w.sum(axis=0)       # w' var  /  1'(W var)
            q0 += M.T @ lw
            const_terms.append(("lin", lw, Mx, Mw, m0))
        else:
            W = np.diag(w[:, 0] ** 2) if wtype == "vector" else w      # ||w o var||^2 / var' W var
            Ws = W + W.T                                               # d/dvar (var' W var) = (W+W') var
            P += M.T @ Ws @ M
            q0 += (M.T @ Ws @ m0)[:, 0]
            Qx += M.T @ Ws @ Mx
            Qw += M.T @ Ws @ Mw
            const_terms.append(("quad", W, Mx, Mw, m0))
    return dict(P=P, q0=q0, Qx=Qx, Qw=Qw, const_terms=const_terms)


def cost_const(const_terms, x0, omega):
    """Constant part r(x_k, omega) of the objective (value when v = 0)."""
    x0 = np.asarray(x0, float).reshape(-1, 1)
    omega = np.asarray(omega, float).reshape(-1, 1)
    r = 0.0
    for kind, w, Mx, Mw, m0 in const_terms:
        c = Mx @ x0 + Mw @ omega + m0
        if kind == "lin":
            r += float(w @ c[:, 0])
        else:
            r += float(c[:, 0] @ w @ c[:, 0])
    return r


def standard_form(mats, atoms, N_p, N_tilde, nu_l=0, nmu_l=0):
    """Everything the solve needs, for one model:  min 1/2 v'Pv + q(x,w)'v + r  s.t.  G v <= h(x,w),
    mu >= 0, v_i in {0,1} (i in bin).  Constraint form: controllers/controller_base.py:446-452."""
    evo = condense_tv(mats, N_tilde) if isinstance(mats, (list, tuple)) else condense(mats, N_tilde)   # list = one model per step
    d = evo["dims"]
    weights = build_weights(atoms, d, N_p, N_tilde)
    cost = assemble_cost(weights, evo, d, N_tilde)
    nv = d["nv"]
    step_bin = var_types(d, nu_l=nu_l, nmu_l=nmu_l)
    is_bin = np.tile(step_bin, N_tilde)
    lb = np.full(N_tilde * nv, -np.inf)
    ub = np.full(N_tilde * nv, np.inf)
    mu_off = d["nu"] + d["ndelta"] + d["nz"]
    for k in range(N_tilde):
        lb[k * nv + mu_off:(k + 1) * nv] = 0.0                    # mu >= 0  (variables.py:221)
    lb[is_bin] = 0.0
    ub[is_bin] = 1.0
    return dict(evo=evo, dims=d, cost=cost, G=evo["H_v"], is_bin=is_bin, lb=lb, ub=ub, weights=weights)


def rhs(evo, x0, omega):
    """h = H_x x_k + H_omega omega + H_5 -- controllers/controller_base.py:446-450."""
    x0 = np.asarray(x0, float).reshape(-1, 1)
    omega = np.asarray(omega, float).reshape(-1, 1)
    return (evo["H_x"] @ x0 + evo["H_omega"] @ omega + evo["H_5"])[:, 0]


def rhs_scenarios(evo, x0, omega_scenarios):
    """Scenario form: row-min over scenario columns of H_omega @ Omega -- controller_base.py:442-444."""
    x0 = np.asarray(x0, float).reshape(-1, 1)
    Hw = np.min(evo["H_omega"] @ np.asarray(omega_scenarios, float), axis=1, keepdims=True)
    return (evo["H_x"] @ x0 + Hw + evo["H_5"])[:, 0]


def lin_cost(cost, x0, omega):
    x0 = np.asarray(x0, float).reshape(-1)
    omega = np.asarray(omega, float).reshape(-1)
    return cost["q0"] + cost["Qx"] @ x0 + cost["Qw"] @ omega
