"""Host-side mirror of the reference's objective atoms (controllers/components/objective_atoms.py).

Only what the MPC hot path needs: parsing the string keys
``<q|Q>[_Linear|_Quadratic|_L1|_L22|_Linf]_[d]<var>[_N_tilde|_N_p|_f]`` (:453-496), tiling per-step
weights over the horizon (VectorWeight :76-137, MatrixWeight :140-206), dropping all-zero weights
(:508,519-520), and turning Linear / Quadratic atoms (:308-331) into the tiled linear / quadratic
weight arrays that libmldgpu's cost pull-back kernel (K4) consumes.  L1 / Linf atoms become epigraph rows of an
augmented MLD model (epigraph.py), rate atoms ordinary atoms on rate outputs of a model with lag states.  This is string handling and
array tiling -- no hot-path arithmetic happens here.
"""
import re

import numpy as np

VAR_NAMES = ("x", "u", "delta", "z", "omega", "y", "mu", "v")
_ATOM_PAT = re.compile(r"(Linear)|(Quadratic)|([L](1|(22)|(inf)))")
_RATE_PAT = re.compile(r"[dD][^e]")       # 'd<var>' marks a rate atom; [^e] keeps 'delta' a variable name
_POSTFIX = ("N_p", "N_tilde", "f", "")


def atleast_2d_col(a):
    """utils/matrix_utils.py:31-39"""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 0:
        return a.reshape(1, 1)
    if a.ndim == 1:
        return a[:, np.newaxis]
    return a


def is_scalar_like(val):
    """utils/matrix_utils.py:20-22: every dimension is 1 (objects without a shape count as scalars)"""
    return all(dd == 1 for dd in getattr(val, "shape", (1,)))


def matmul(a, b):
    """utils/matrix_utils.py:24-28: elementwise product when either operand is scalar-like (1 x 1 matrices broadcast), the
    matrix product otherwise -- the rule every `H_x x_k`, `Phi_x x_k`, `C x` of the reference goes through"""
    if is_scalar_like(a) or is_scalar_like(b):
        return a * b
    return a @ b


def parse_key(key):
    info = key.split("_")
    weight_type = "vector" if "".join(info[0:1]).islower() else "matrix"
    atom_type = "".join(info[1:2]).capitalize()
    if not _ATOM_PAT.search(atom_type):
        atom_type = "Linear" if weight_type == "vector" else "Quadratic"
        var = "".join(info[1:2]).lower()
        post = "_".join(info[2:])
    else:
        var = "".join(info[2:3]).lower()
        post = "_".join(info[3:])
    rate = False
    if _RATE_PAT.search(var):
        var, rate = var[1:], True
    if var not in VAR_NAMES or post not in _POSTFIX:
        raise ValueError(
            "weight_name: '%s' is not valid. Must be of the form:\n"
            "  \"lower/upper[_Linear|_Quadratic|_L1|_L22|_Linf]_[d]var_name[_N_tilde|_N_p|_f]\"" % key)
    if atom_type == "L22":
        atom_type = "Quadratic"
    return weight_type, atom_type, var, rate, post


def _block_diag(block, k):
    r, c = block.shape
    out = np.zeros((k * r, k * c))
    for i in range(k):
        out[i * r:(i + 1) * r, i * c:(i + 1) * c] = block
    return out


def _tile(value, var_dim, length, weight_type, var, terminal=False):
    value = atleast_2d_col(value)
    if weight_type == "vector":
        if value.shape[1] != 1:
            raise ValueError("Column dim of vector weight for opt_var: '%s', must be 1." % var)
        if terminal:
            if value.shape[0] != var_dim:
                raise ValueError("Row dim of vector terminal weight for opt_var: '%s' must be in {%d}" % (var, var_dim))
            return value
        if value.shape[0] == var_dim * length:
            return value
        if value.shape[0] == var_dim:
            return np.tile(value, (length, 1))
        raise ValueError("Row dim of vector weight for opt_var: '%s', must be in {%d, %d*%d}" % (var, var_dim, var_dim, length))
    if value.shape[0] != value.shape[1]:
        raise ValueError("matrix weight for opt_var: '%s', must be square. Currently has shape: %s" % (var, value.shape))
    if terminal:
        if value.shape[0] != var_dim:
            raise ValueError("Row dim of matrix terminal weight for opt_var: '%s' must be in {%d}" % (var, var_dim))
        return value
    if value.shape[0] == var_dim * length:
        return value
    if value.shape[0] == var_dim:
        return _block_diag(value, length)
    raise ValueError("Row dim of matrix weight for opt_var: '%s', must be in {%d, %d*%d}" % (var, var_dim, var_dim, length))


class ObjectiveAtoms(object):
    """dict-like: (var, atom_type, weight_type, is_rate) -> weight_N_tilde ndarray"""

    def __init__(self, dims, N_p, N_tilde, objective_atoms_struct=None, **kwargs):
        self.dims = dims
        self.N_p, self.N_tilde = int(N_p), int(N_tilde)
        self.weights = {}
        self.update(objective_atoms_struct, **kwargs)

    def var_dim(self, var):
        d = self.dims
        return dict(x=d["nx"], u=d["nu"], delta=d["ndelta"], z=d["nz"], omega=d["nomega"], y=d["ny"], mu=d["nmu"],
                    v=d["nu"] + d["ndelta"] + d["nz"] + d["nmu"])[var]

    def set(self, objective_atoms_struct=None, **kwargs):
        self.weights = {}
        self.update(objective_atoms_struct, **kwargs)

    def update(self, objective_atoms_struct=None, **kwargs):
        struct = dict(objective_atoms_struct or {})
        struct.update(kwargs)
        N_p, N_t = self.N_p, self.N_tilde
        for key, value in struct.items():
            if value is None:
                continue
            wtype, atype, var, rate, post = parse_key(key)
            value = atleast_2d_col(value)
            vd = self.var_dim(var)
            if post:
                which = post
            elif value.shape[0] == vd or value.shape[0] == vd * N_t:
                which = "N_tilde"
            else:
                which = "N_p"
            ident = (var, atype, wtype, rate)
            w = self.weights.get(ident)
            if w is None:
                if np.all(np.isclose(value, 0.0)):
                    continue
                w = np.zeros((N_t * vd, 1)) if wtype == "vector" else np.zeros((N_t * vd, N_t * vd))
            if which == "N_p" and N_p > N_t:
                raise ValueError("Cannot set weight_N_p if N_tilde < N_p")
            if wtype == "vector":
                if which == "N_tilde":
                    w[:] = _tile(value, vd, N_t, wtype, var)
                elif which == "N_p":
                    w[:N_p * vd, :1] = _tile(value, vd, N_p, wtype, var)
                else:
                    w[-vd:, :1] = _tile(value, vd, 1, wtype, var, terminal=True)
            else:
                if which == "N_tilde":
                    w[:] = _tile(value, vd, N_t, wtype, var)
                elif which == "N_p":
                    w[:N_p * vd, :N_p * vd] = _tile(value, vd, N_p, wtype, var)
                else:
                    w[-vd:, -vd:] = _tile(value, vd, 1, wtype, var, terminal=True)
            if np.all(np.isclose(w, 0.0)):
                self.weights.pop(ident, None)
            else:
                self.weights[ident] = w

    # ------------------------------------------------------------------------------------------
    def _v_rows(self, var):
        d, N = self.dims, self.N_tilde
        nv = self.var_dim("v")
        offs = dict(u=(0, d["nu"]), delta=(d["nu"], d["ndelta"]), z=(d["nu"] + d["ndelta"], d["nz"]),
                    mu=(d["nu"] + d["ndelta"] + d["nz"], d["nmu"]), v=(0, nv))
        o, k = offs[var]
        if not k:
            return np.zeros(0, dtype=int)
        return np.concatenate([np.arange(s * nv + o, s * nv + o + k) for s in range(N)])

    def to_cost(self):
        """tiled weights for libmldgpu (include/mldgpu.h mld_cost).  v_tilde = [u0;d0;z0;mu0;u1;...]
        (controllers/components/variables.py:226-241)."""
        d, N = self.dims, self.N_tilde
        n = N * self.var_dim("v")
        lin = dict(v=np.zeros(n), x=np.zeros(N * d["nx"]), y=np.zeros(N * d["ny"]))
        quad = dict(v=None, x=None, y=None)
        const_omega = []
        for (var, atype, wtype, rate), w in self.weights.items():
            if rate:                         # rate atoms live on the lag-state augmentation (epigraph.augment_rates)
                continue
            if atype in ("L1", "Linf"):      # epigraph atoms: see epigraph_blocks() / epigraph.py (a model augmentation)
                if var == "omega":
                    raise NotImplementedError("%s atom on omega" % atype)
                continue
            if atype == "L22":               # L22ObjectiveAtom subclasses the quadratic atom (objective_atoms.py:347-348)
                atype = "Quadratic"
            if atype == "Linear":            # w' var  /  sum(W var)          (objective_atoms.py:314-318)
                lw = w[:, 0] if wtype == "vector" else w.sum(axis=0)
                W = None
            else:                            # ||w o var||^2 / var' W var      (:327-331)
                W = np.diag(w[:, 0] ** 2) if wtype == "vector" else w
                lw = None
            if var in ("x", "y"):
                tgt, rows = var, slice(None)
            elif var == "omega":
                const_omega.append((atype, lw, W))
                continue
            else:
                tgt, rows = "v", self._v_rows(var)
            if lw is not None:
                if tgt == "v":
                    lin["v"][rows] += lw
                else:
                    lin[tgt] += lw
            else:
                size = lin[tgt].size
                if quad[tgt] is None:
                    quad[tgt] = np.zeros((size, size))
                if tgt == "v":
                    quad["v"][np.ix_(rows, rows)] += W
                else:
                    quad[tgt] += W
        out = dict(lin_v=lin["v"], lin_x=lin["x"], lin_y=lin["y"], quad_v=quad["v"], quad_x=quad["x"], quad_y=quad["y"])
        out["_omega_atoms"] = const_omega
        return out

    def epigraph_blocks(self):
        """epigraph rows / auxiliaries the L1 and Linf atoms need (empty list when there are none)"""
        from . import epigraph
        return epigraph.plan(self.weights, self.dims, self.N_tilde)
