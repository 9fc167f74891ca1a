"""Per-step auxiliary resolution on the GPU: the (delta, z, mu) that go with a given (x, u, omega).

Replaces ``MldModel._compute_aux`` (models/mld_model.py:701-766), which the reference calls from ``lsim_k``
(:647-699) for every device at every simulation step (``ControllerBase.sim_step_k``, controller_base.py:229-253)
and solves as a cvxpy/Gurobi feasibility MIP

    find delta in {0,1}, z, mu >= 0, y :   y = C x + D1 u + D2 delta + D3 z + D4 omega + d5,
                                           E x + F1 u + F2 delta + F3 z + F4 omega + G y + Psi mu <= f5 .

Here the same problem is the horizon-1 (N_tilde = 1) instance of the batched GPU path: everything that is known
(u and whichever of delta / z / mu the caller supplies) is folded into the disturbance channel, omega' =
[omega; u; known...], so that the standard kernels (condense -> rhs -> cut-and-branch) solve for the unknown
block only, one instance per (x, u, omega) triple and thousands of triples per launch.

The reference minimises the constant 0, i.e. returns *some* feasible point; this implementation returns the
feasible point with the smallest total slack ``sum(mu)`` (and, with mu known or absent, any vertex), which is one
of the points the reference may return and makes the result deterministic.  No CPU fallback.
"""
import numpy as np

from . import gpu, host

_AUX = ("delta", "z", "mu")
_COLS = dict(u=("B1", "D1", "F1"), delta=("B2", "D2", "F2"), z=("B3", "D3", "F3"), mu=(None, None, "Psi"))


def fold_known(mats, dims, unknown):
    """(mats', dims') of the MLD system whose inputs are only the ``unknown`` auxiliaries; u and the known
    auxiliaries become extra columns of the disturbance matrices B4 / D4 / F4 (in the order u, delta, z, mu)."""
    nx, ny, nc = dims["nx"], dims["ny"], dims["nc"]
    rows = dict(B=nx, D=ny, F=nc)

    def get(name, r, c):
        a = mats.get(name) if name else None
        if a is None or np.size(a) == 0:
            return np.zeros((r, c))
        return np.asarray(a, dtype=np.float64).reshape(r, c)

    known = ["u"] + [v for v in _AUX if v not in unknown]
    out = {k: mats.get(k) for k in ("A", "b5", "C", "d5", "E", "f5", "G")}
    b4 = [get("B4", nx, dims["nomega"])]
    d4 = [get("D4", ny, dims["nomega"])]
    f4 = [get("F4", nc, dims["nomega"])]
    for v in known:
        nb, nd, nf = _COLS[v]
        w = dims["n" + v]
        b4.append(get(nb, nx, w)); d4.append(get(nd, ny, w)); f4.append(get(nf, nc, w))
    out["B4"], out["D4"], out["F4"] = np.hstack(b4), np.hstack(d4), np.hstack(f4)
    d2 = dict(nx=nx, ny=ny, nc=nc, nu=0, nu_l=0, nomega=out["B4"].shape[1],
              ndelta=dims["ndelta"] if "delta" in unknown else 0, nz=dims["nz"] if "z" in unknown else 0,
              nmu=dims["nmu"] if "mu" in unknown else 0, nmu_l=dims.get("nmu_l", 0) if "mu" in unknown else 0)
    for v in _AUX:
        nb, nd, nf = _COLS[v]
        if v in unknown:
            if nb:
                out[nb], out[nd] = mats.get(nb), mats.get(nd)
            out[nf] = mats.get(nf)
    return out, d2, known


class AuxResolver(object):
    """GPU problem for one numeric MLD model and one set of unknown auxiliaries."""

    def __init__(self, mats, dims, unknown=_AUX, **solver_opts):
        self.dims = dict(dims)
        self.unknown = tuple(v for v in _AUX if v in unknown and dims["n" + v] > 0)
        self.mats2, self.dims2, self.known = fold_known(mats, dims, self.unknown)
        self.nv2 = self.dims2["ndelta"] + self.dims2["nz"] + self.dims2["nmu"]
        self._model = self._problem = None
        if self.nv2:
            atoms = {"q_mu": np.ones((self.dims2["nmu"], 1))} if self.dims2["nmu"] else {}
            opts = dict(gap_abs=1e-9, gap_rel=0.0, max_nodes=20000)
            opts.update(solver_opts)
            self._model = gpu.GpuModel([self.mats2], self.dims2)
            self._problem = gpu.GpuProblem(self._model, 0, 1, host.cost_from_atoms(atoms, self.dims2, 0, 1), **opts)

    def resolve(self, x, u, omega, delta=None, z=None, mu=None):
        """batched: x (B,nx), u (B,nu), omega (B,nomega) and the known auxiliaries -> dict(delta, z, mu, status).
        Rows whose problem is infeasible come back as NaN, as the reference does (:757-763)."""
        d = self.dims
        x = np.asarray(x, np.float64).reshape(-1, d["nx"])
        B = x.shape[0]
        given = dict(u=u, delta=delta, z=z, mu=mu)
        parts = [np.asarray(omega, np.float64).reshape(B, d["nomega"])]
        for v in self.known:
            a = given[v]
            parts.append(np.zeros((B, d["n" + v])) if a is None else np.asarray(a, np.float64).reshape(B, d["n" + v]))
        w2 = np.ascontiguousarray(np.hstack(parts))
        out = {v: (None if given[v] is None else np.asarray(given[v], np.float64).reshape(B, d["n" + v])) for v in _AUX}
        status = np.zeros(B, np.int32)
        if self.nv2:
            r = self._problem.solve(np.ascontiguousarray(x), w2)
            status = r["status"]
            v = r["v"]
            v[~np.isfinite(r["obj"])] = np.nan
            o = 0
            for name in _AUX:
                if name in self.unknown:
                    w = d["n" + name]
                    out[name] = v[:, o:o + w]
                    o += w
        for name in _AUX:
            if out[name] is None:
                out[name] = np.zeros((B, d["n" + name]))
        out["status"] = status
        return out

    def close(self):
        if self._problem is not None:
            self._problem.close(); self._problem = None
        if self._model is not None:
            self._model.close(); self._model = None
