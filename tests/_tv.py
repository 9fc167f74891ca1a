"""Time-varying horizons for the tests: per-step variations of a synthetic agent's MLD model."""
import numpy as np


def step_models(mats, N, seed=0, strength=0.05):
    """N step models: the dynamics (A, B1, B4, b5), the output map (C, d5) and the constraint offsets f5 drift with the
    step index.  f5 only grows (rows only loosen), so a feasible time-invariant instance stays feasible."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(N):
        m = {key: (None if val is None else np.array(val, dtype=np.float64, copy=True)) for key, val in mats.items()}
        for key in ("A", "B1", "B4", "b5", "C", "d5"):
            a = m.get(key)
            if a is not None and a.size:
                m[key] = a * (1.0 + strength * rng.uniform(-1.0, 1.0, a.shape))
        f5 = m.get("f5")
        if f5 is not None and f5.size:
            m["f5"] = f5 + strength * rng.uniform(0.0, 1.0, f5.shape) * np.maximum(1.0, np.abs(f5))
        out.append(m)
    return out


def simulate(ms, d, x0, V, W):
    """step-by-step evolution with one model per step: stacked x(k), y(k) and row residuals E x + F v + F4 w + G y - f5"""
    def g(m, name, r, c):
        a = m.get(name)
        return np.zeros((r, c)) if a is None or np.size(a) == 0 else np.asarray(a, np.float64).reshape(r, c)
    nx, ny, nc, nw = d["nx"], d["ny"], d["nc"], d["nomega"]
    widths = (d["nu"], d["ndelta"], d["nz"], d["nmu"])
    x = x0.reshape(nx, 1)
    xs, ys, res = [], [], []
    for k, m in enumerate(ms):
        v, w = V[k].reshape(-1, 1), W[k].reshape(-1, 1)
        o = np.cumsum((0,) + widths)
        u, de, z, mu = (v[o[i]:o[i + 1]] for i in range(4))
        y = (g(m, "C", ny, nx) @ x + g(m, "D1", ny, widths[0]) @ u + g(m, "D2", ny, widths[1]) @ de + g(m, "D3", ny, widths[2]) @ z
             + g(m, "D4", ny, nw) @ w + g(m, "d5", ny, 1))
        r = (g(m, "E", nc, nx) @ x + g(m, "F1", nc, widths[0]) @ u + g(m, "F2", nc, widths[1]) @ de + g(m, "F3", nc, widths[2]) @ z
             + g(m, "F4", nc, nw) @ w + g(m, "G", nc, ny) @ y + g(m, "Psi", nc, widths[3]) @ mu - g(m, "f5", nc, 1))
        xs.append(x); ys.append(y); res.append(r)
        x = (g(m, "A", nx, nx) @ x + g(m, "B1", nx, widths[0]) @ u + g(m, "B2", nx, widths[1]) @ de + g(m, "B3", nx, widths[2]) @ z
             + g(m, "B4", nx, nw) @ w + g(m, "b5", nx, 1))
    return np.vstack(xs), np.vstack(ys), np.vstack(res)
