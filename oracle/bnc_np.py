"""ORACLE (test infrastructure, never shipped): numpy prototype of the exact MILP algorithm.

Dense-dictionary bounded dual simplex + Gomory mixed-integer (GMI) cut rounds at the root +
iterative-deepening depth-first branch-and-bound that re-uses ONE dictionary per instance
(branching = bound changes; every binary is boxed, so any basis can be made dual feasible by
bound flips).

The reference delegates this step to cvxpy -> Gurobi (controllers/controller_base.py:509), a
third-party proprietary branch-and-cut that is not in /root/reference and not installable here; this
file restates the *published* textbook algorithm (bounded dual simplex with Harris ratio test:
Chvatal 1983 ch. 8/10, Harris 1973; GMI cuts: Gomory 1960 / Balas-Ceria-Cornuejols-Natraj 1996;
LP-based branch-and-bound: Land-Doig 1960; depth-first iterative deepening: Korf 1985) for the
standard form built from controller_base.py:446-452 and variables.py:189-243.
Slow (pure numpy): used for small cases and to cross-check oracle/mld_oracle.c.
"""
import numpy as np

BIG = 1.0e7        # artificial box for a free variable that must start at a bound
PTOL = 1e-8        # primal feasibility tolerance (scaled space)
PTOL_SKIP = 1e-6   # a row violated by less than this with no usable pivot is treated as feasible
DTOL = 1e-9        # Harris dual tolerance
PIV_ABS = 1e-7     # minimum |pivot| (scaled space)
PIV_REL = 1e-7     # ... and relative to the largest eligible entry of the row
INTTOL = 1e-6
COEF_ZERO = 1e-9   # tableau entries below this are treated as exact zeros when cutting
RESID_TOL = 1e-6   # dictionary-vs-original discrepancy that triggers a refactor


def equilibrate(G, is_int, passes=3):
    """row scales (all rows) and column scales (continuous columns only) so that entries are O(1)"""
    m, n = G.shape
    rs = np.ones(m)
    cs = np.ones(n)
    A = np.abs(G)
    for _ in range(passes):
        rmax = (A * rs[:, None] * cs[None, :]).max(axis=1)
        rs = np.where(rmax > 0, rs / np.where(rmax > 0, rmax, 1.0), rs)
        cmax = (A * rs[:, None] * cs[None, :]).max(axis=0)
        upd = (~is_int) & (cmax > 0)
        cs = np.where(upd, cs / np.where(cmax > 0, cmax, 1.0), cs)
    # powers of two keep the scaling exact in floating point
    rs = 2.0 ** np.round(np.log2(rs))
    cs = 2.0 ** np.round(np.log2(cs))
    return rs, cs


def propagate_bounds(G, h, lb, ub, is_int, passes=4):
    """row-activity bound propagation with integer rounding (Savelsbergh 1994, section 1.1):
    a_j x_j <= h - min-activity(rest).  Returns tightened (lb, ub) or None if a row is infeasible."""
    lb, ub = lb.copy(), ub.copy()
    m, n = G.shape
    rows = [np.nonzero(G[i])[0] for i in range(m)]
    for _ in range(passes):
        changed = False
        for i in range(m):
            js = rows[i]
            if js.size == 0:
                if h[i] < -1e-9:
                    return None
                continue
            a = G[i, js]
            lo_c = np.where(a > 0, a * lb[js], a * ub[js])      # minimal contribution of each term
            ninf = np.isinf(lo_c)
            tot = lo_c[~ninf].sum()
            k = int(ninf.sum())
            if k == 0 and tot > h[i] + 1e-7 * max(1.0, abs(h[i])):
                return None
            for t, j in enumerate(js):
                if k == 0:
                    rest = tot - lo_c[t]
                elif k == 1 and ninf[t]:
                    rest = tot
                else:
                    continue
                bnd = (h[i] - rest) / a[t]
                if a[t] > 0:
                    if is_int[j]:
                        bnd = np.floor(bnd + 1e-7)
                    if bnd < ub[j] - 1e-9 * max(1.0, abs(bnd)):
                        ub[j] = bnd
                        changed = True
                else:
                    if is_int[j]:
                        bnd = np.ceil(bnd - 1e-7)
                    if bnd > lb[j] + 1e-9 * max(1.0, abs(bnd)):
                        lb[j] = bnd
                        changed = True
        if np.any(lb > ub + 1e-7):
            return None
        if not changed:
            break
    return lb, np.maximum(ub, lb)


class Dictionary(object):
    """Scaled problem  min q'x  s.t.  Gx + s = h, s >= 0, lo <= x <= hi  held as
    xB[r] = beta[r] - sum_c D[r,c] * xN[c];  objective = z0 + sum_c d[c]*xN[c]."""

    def __init__(self, q, G, h, lb, ub, is_int, max_cuts=0):
        m, n = G.shape
        self.n, self.m0, self.mcap = n, m, m + max_cuts
        self.is_int = np.asarray(is_int, bool)
        self.rs, self.cs = equilibrate(G, self.is_int)
        self.Gx = np.zeros((self.mcap, n))            # extended constraint rows (original + cuts), scaled
        self.Gx[:m] = G * self.rs[:, None] * self.cs[None, :]
        self.hx = np.zeros(self.mcap)
        self.hx[:m] = h * self.rs
        self.q = q * self.cs
        self.ntot = n + self.mcap                      # ids: 0..n-1 structural, n+i slack of row i
        self.lo = np.concatenate([lb / self.cs, np.zeros(self.mcap)])
        self.hi = np.concatenate([ub / self.cs, np.full(self.mcap, np.inf)])
        self.m = m
        self.pivots = 0
        self.refactors = 0
        self.basic = np.arange(n, n + self.mcap).astype(int)
        self.nonbasic = np.arange(n).astype(int)
        self.at_upper = np.zeros(n, dtype=bool)
        self.xN = np.zeros(n)
        self._reset_dictionary()
        for c in range(n):
            self._place(c)
        self.refresh()

    # ------------------------------------------------------------------ basics
    def _reset_dictionary(self):
        self.D = self.Gx.copy()
        self.beta = self.hx.copy()
        self.d = self.q.copy()
        self.z0 = 0.0

    def _place(self, c):
        """put nonbasic column c at the bound that makes it dual feasible (boxing a free side)"""
        j = self.nonbasic[c]
        if self.lo[j] == self.hi[j]:
            self.at_upper[c], self.xN[c] = False, self.lo[j]
        elif self.d[c] >= 0:
            if not np.isfinite(self.lo[j]):
                self.lo[j] = -BIG
            self.at_upper[c], self.xN[c] = False, self.lo[j]
        else:
            if not np.isfinite(self.hi[j]):
                self.hi[j] = BIG
            self.at_upper[c], self.xN[c] = True, self.hi[j]

    def refresh(self):
        self.xB = self.beta[:self.m] - self.D[:self.m] @ self.xN

    def objective(self):
        return self.z0 + float(self.d @ self.xN)

    def values(self):
        x = np.zeros(self.ntot)
        x[self.nonbasic] = self.xN
        x[self.basic[:self.m]] = self.xB
        return x

    def x_original(self):
        return self.values()[:self.n] * self.cs

    def set_bounds(self, j, lo, hi):
        """change bounds of variable j (branching / backtracking) keeping dual feasibility"""
        self.lo[j], self.hi[j] = lo, hi
        cs = np.where(self.nonbasic == j)[0]
        if cs.size:
            c = cs[0]
            old = self.xN[c]
            if lo == hi:
                new, self.at_upper[c] = lo, False
            elif self.d[c] >= 0:
                new, self.at_upper[c] = lo, False
            else:
                new, self.at_upper[c] = hi, True
            if new != old:
                self.xB -= self.D[:self.m, c] * (new - old)
                self.xN[c] = new

    # ------------------------------------------------------------------ dual simplex
    def dual_simplex(self, cutoff=np.inf, max_iter=20000):
        """'optimal' | 'infeasible' | 'cutoff' | 'iterlimit'"""
        m = self.m
        skip = np.zeros(m, dtype=bool)
        stall, last_obj = 0, -np.inf
        for _ in range(max_iter):
            cur = self.objective()
            if cur > last_obj + 1e-12 * max(1.0, abs(cur)):
                stall, last_obj = 0, cur
            else:
                stall += 1
            bland = stall > 30      # anti-cycling: Bland's smallest-index rules while stalling
            lo = self.lo[self.basic[:m]]
            hi = self.hi[self.basic[:m]]
            viol_lo = lo - self.xB
            viol_hi = self.xB - hi
            viol = np.where(skip, -np.inf, np.maximum(viol_lo, viol_hi))
            r = int(np.argmax(viol))
            if viol[r] <= PTOL:
                if self.check_residual() > RESID_TOL:
                    self.refactor()
                    skip[:] = False
                    continue
                return "optimal"
            if bland:
                ids = np.where(viol > PTOL, self.basic[:m], np.iinfo(np.int64).max)
                r = int(np.argmin(ids))
            if cur >= cutoff:
                return "cutoff"
            below = viol_lo[r] > viol_hi[r]
            row = self.D[r]
            fixed = self.lo[self.nonbasic] == self.hi[self.nonbasic]
            # below: xB must go up -> at-lower columns with row<0, at-upper columns with row>0
            if below:
                elig = np.where(self.at_upper, row > 0, row < 0) & ~fixed
            else:
                elig = np.where(self.at_upper, row < 0, row > 0) & ~fixed
            arow = np.abs(row)
            ptol = max(PIV_ABS, PIV_REL * float(arow[elig].max())) if elig.any() else PIV_ABS
            cand = elig & (arow > ptol)
            if not cand.any():
                if viol[r] <= PTOL_SKIP:
                    skip[r] = True
                    continue
                return "infeasible"
            idx = np.where(cand)[0]
            dabs = np.maximum(np.where(self.at_upper[idx], -self.d[idx], self.d[idx]), 0.0)
            ar = arow[idx]
            if bland:
                ratios = dabs / ar
                tie = idx[ratios <= ratios.min() * (1 + 1e-12) + 1e-300]
                c = int(tie[np.argmin(self.nonbasic[tie])])
            else:
                # Harris two-pass ratio test: relaxed bound on the step, then the largest pivot
                tmax = np.min((dabs + DTOL) / ar)
                sub = idx[dabs / ar <= tmax]
                c = int(sub[np.argmax(arow[sub])])
            self.pivot(r, c, lo[r] if below else hi[r])
        return "iterlimit"

    def pivot(self, r, c, leave_value):
        m = self.m
        D, beta, d = self.D, self.beta, self.d
        p = D[r, c]
        theta = (self.xB[r] - leave_value) / p          # change of the entering variable
        colc = D[:m, c].copy()
        self.xB -= colc * theta
        enter_val = self.xN[c] + theta
        rowr = D[r] / p
        br = beta[r] / p
        rowr[c] = 1.0 / p
        mult = colc
        mult[r] = 0.0
        D[:m] -= np.outer(mult, rowr)
        D[:m, c] = -mult / p
        beta[:m] -= mult * br
        D[r] = rowr
        beta[r] = br
        dc = d[c]
        self.z0 += dc * br
        d -= dc * rowr
        d[c] = -dc / p
        jb, jn = self.basic[r], self.nonbasic[c]
        self.basic[r], self.nonbasic[c] = jn, jb
        self.xB[r] = enter_val
        self.xN[c] = leave_value
        self.at_upper[c] = (leave_value == self.hi[jb]) and (self.lo[jb] != self.hi[jb])
        self.pivots += 1

    # ------------------------------------------------------------------ verification / refactor
    def check_residual(self):
        """largest discrepancy between the dictionary's slack values and h - Gx from original data"""
        if not self.m:
            return 0.0
        v = self.values()
        true_s = self.hx[:self.m] - self.Gx[:self.m] @ v[:self.n]
        return float(np.max(np.abs(true_s - v[self.n:self.n + self.m])))

    def refactor(self):
        """rebuild the dictionary for the current basis from the original rows (complete pivoting)"""
        self.refactors += 1
        m, n = self.m, self.n
        want_basic = [j for j in self.basic[:m] if j < n]            # structurals that must be basic
        state = {j: (self.at_upper[c], self.xN[c]) for c, j in enumerate(self.nonbasic)}
        self._reset_dictionary()
        self.basic = np.arange(n, n + self.mcap).astype(int)
        self.nonbasic = np.arange(n).astype(int)
        rows_free = np.array([(n + i) in state for i in range(m)])   # slacks that must become nonbasic
        cols_free = np.zeros(n, dtype=bool)
        cols_free[want_basic] = True
        self.xN = np.zeros(n)
        self.xB = np.zeros(m)
        saved_piv = self.pivots
        for _ in range(len(want_basic)):
            sub = np.abs(self.D[:m][np.ix_(rows_free, cols_free)])
            k = int(np.argmax(sub))
            ri, ci = np.unravel_index(k, sub.shape)
            r = np.where(rows_free)[0][ri]
            c = np.where(cols_free)[0][ci]
            self.pivot(r, c, 0.0)
            rows_free[r] = False
            cols_free[c] = False
        self.pivots = saved_piv
        for c, j in enumerate(self.nonbasic):
            self.at_upper[c], self.xN[c] = state[j]
        self.refresh()

    # ------------------------------------------------------------------ GMI cuts
    def gmi_round(self, max_cuts, min_frac=1e-3, max_dyn=1e6):
        """append GMI cuts (stored as ordinary rows in structural space) for fractional integer basics"""
        m, n = self.m, self.n
        nb = self.nonbasic
        fr = []
        for r in range(m):
            j = self.basic[r]
            if j < n and self.is_int[j]:
                f0 = self.xB[r] - np.floor(self.xB[r])
                if min_frac < f0 < 1 - min_frac:
                    fr.append((abs(f0 - 0.5), r))
        fr.sort()
        nb_int = (nb < n) & self.is_int[np.minimum(nb, n - 1)]
        fixed = self.lo[nb] == self.hi[nb]
        sign = np.where(self.at_upper, -1.0, 1.0)
        v = self.values()
        new_rows = []
        for _, r in fr:
            if self.m + len(new_rows) >= self.mcap or len(new_rows) >= max_cuts:
                break
            f0 = self.xB[r] - np.floor(self.xB[r])
            a = np.where(fixed, 0.0, self.D[r] * sign)      # xB + sum_c a_c t_c = value, t_c >= 0
            a = np.where(np.abs(a) < COEF_ZERO, 0.0, a)     # tableau noise
            fj = a - np.floor(a)
            fj = np.where((fj < COEF_ZERO) | (fj > 1 - COEF_ZERO), 0.0, fj)
            gi = np.where(fj <= f0, fj / f0, (1 - fj) / (1 - f0))
            gc = np.where(a > 0, a / f0, -a / (1 - f0))
            g = np.where(nb_int, gi, gc)
            nz = g[g > 0]
            if nz.size == 0 or nz.max() / nz.min() > max_dyn:
                continue
            # cut  sum_c g_c t_c >= 1  rewritten over structural variables:  ax . x <= bx
            ax = np.zeros(n)
            bx = -1.0
            for c in np.where(g > 0)[0]:
                j = nb[c]
                if j < n:      # t = sign*(x_j - bound)
                    bnd = self.hi[j] if self.at_upper[c] else self.lo[j]
                    ax[j] -= g[c] * sign[c]
                    bx -= g[c] * sign[c] * bnd
                else:          # slack of row i: t = s_i = hx_i - Gx_i x
                    i = j - n
                    ax += g[c] * self.Gx[i]
                    bx += g[c] * self.hx[i]
            nrm = np.abs(ax).max()
            if nrm <= 0:
                continue
            new_rows.append((ax / nrm, bx / nrm, g / nrm))
        for ax, bx, g in new_rows:
            k = self.m
            self.Gx[k] = ax
            self.hx[k] = bx
            # slack of the new row in the current dictionary: s = (sum_c g_c t_c - 1)/nrm, i.e.
            # s = beta_k - D_k xN with D_k = -(g*sign); beta_k from the current point
            self.D[k] = -(g * sign)
            self.beta[k] = (bx - ax @ v[:n]) + self.D[k] @ self.xN
            self.basic[k] = n + k
            self.m += 1
        if new_rows:
            self.refresh()
        return len(new_rows)


def solve_milp(q, G, h, lb, ub, is_bin, gap_abs=1e-9, gap_rel=0.0, max_nodes=100000, cut_rounds=8,
               cuts_per_round=40, max_cuts=200, prio=None, log=None, ids_mode="dfs", presolve=True):
    """Exact MILP by cut-and-branch. Returns dict(status, obj, x, nodes, pivots, ...)."""
    n = q.size
    is_bin = np.asarray(is_bin, bool)
    res = dict(status="infeasible", obj=np.inf, x=None, nodes=0, pivots=0, cuts=0, root_lp=None,
               root_bound=None, passes=0, refactors=0)
    if presolve:
        pb = propagate_bounds(G, h, lb, ub, is_bin)
        if pb is None:
            return res
        lb, ub = pb
    dic = Dictionary(q, G, h, lb, ub, is_bin, max_cuts=max_cuts)
    st = dic.dual_simplex()
    res = dict(status="infeasible", obj=np.inf, x=None, nodes=1, pivots=0, cuts=0, root_lp=None,
               root_bound=None, passes=0, refactors=0)
    if st != "optimal":
        res["pivots"] = dic.pivots
        return res
    res["root_lp"] = dic.objective()
    ncuts = 0
    for rnd in range(cut_rounds):
        before = dic.objective()
        k = dic.gmi_round(cuts_per_round)
        if k == 0:
            break
        ncuts += k
        st = dic.dual_simplex()
        if st != "optimal":
            res.update(status="infeasible", pivots=dic.pivots, cuts=ncuts)
            return res
        if log:
            log("round %d cuts %d bound %.6f -> %.6f" % (rnd, k, before, dic.objective()))
        if dic.objective() - before < 1e-6 * max(1.0, abs(before)):
            break
    res["root_bound"] = dic.objective()
    res["cuts"] = ncuts
    bins = np.where(is_bin)[0]
    if prio is None:
        prio = np.arange(n, dtype=float)
    root_lo, root_hi = dic.lo[:n].copy(), dic.hi[:n].copy()
    best, bestx = np.inf, None
    nodes = 0
    root_bound = res["root_bound"]

    def tol(v):
        return max(gap_abs, gap_rel * abs(v))

    def accept_leaf(x):
        """all binaries within INTTOL: fix them to the rounded values, re-solve, verify, restore"""
        nonlocal best, bestx
        saved = []
        for j in bins:
            if dic.lo[j] != dic.hi[j]:
                saved.append((j, dic.lo[j], dic.hi[j]))
                v = float(np.round(x[j]))
                dic.set_bounds(j, v, v)
        st = dic.dual_simplex()
        if st == "optimal":
            xo = dic.x_original()
            xo[bins] = np.round(xo[bins])
            obj = float(q @ xo)
            if obj < best and np.all((G @ xo - h) * dic.rs <= 1e-6):
                best, bestx = obj, xo
        for j, lo_, hi_ in saved:
            dic.set_bounds(j, lo_, hi_)

    # Depth-first branch-and-bound.  `dive_first`: pass 0 runs with T = +inf until the first
    # incumbent is found (a dive), later passes are iterative deepening on the LP bound: pass k
    # explores every node whose bound is <= min(T_k, incumbent - tol).
    T = np.inf if ids_mode == "dfs" else root_bound + max(1e-7 * max(1.0, abs(root_bound)), tol(root_bound))
    passes = 0
    status = "node_limit"
    while nodes < max_nodes:
        passes += 1
        t_next = [np.inf]
        stack = []

        def evaluate():
            nonlocal nodes
            nodes += 1
            cut = min(T, best - tol(best)) if np.isfinite(best) else T
            st = dic.dual_simplex(cutoff=cut + 1e-12)
            if st == "infeasible":
                return None
            obj = dic.objective()
            if st != "optimal" or obj > cut:
                t_next[0] = min(t_next[0], obj)      # a valid lower bound of this subtree
                return None
            x = dic.values()[:n]
            fr = np.abs(x[bins] - np.round(x[bins]))
            if fr.max() <= INTTOL:
                accept_leaf(x)
                return None
            cand = bins[fr > INTTOL]
            j = cand[np.argmin(prio[cand])]
            return j, x[j]

        out = evaluate()
        while True:
            if np.isfinite(best) and best <= root_bound + tol(best):
                break                # cannot be improved
            if out is not None and nodes < max_nodes:
                j, xj = out
                first = 1.0 if xj >= 0.5 else 0.0
                stack.append([j, first, False])
                dic.set_bounds(j, first, first)
                out = evaluate()
                continue
            while stack and stack[-1][2]:
                j, _, _ = stack.pop()
                dic.set_bounds(j, root_lo[j], root_hi[j])
            if not stack or nodes >= max_nodes:
                break
            stack[-1][2] = True
            j, first = stack[-1][0], stack[-1][1]
            dic.set_bounds(j, 1.0 - first, 1.0 - first)
            out = evaluate()
        exhausted = not stack
        for j, _, _ in stack:
            dic.set_bounds(j, root_lo[j], root_hi[j])
        if np.isfinite(best) and best <= root_bound + tol(best):
            status = "optimal"
            break
        if nodes >= max_nodes:
            break
        # the pass explored every node with bound <= min(T, best - tol)
        if np.isfinite(best) and (best - tol(best) <= T or t_next[0] >= best - tol(best)):
            status = "optimal"
            break
        if not np.isfinite(t_next[0]):
            status = "optimal" if bestx is not None else "infeasible"
            break
        T = max(t_next[0] + 1e-9 * max(1.0, abs(t_next[0])), T + (2.0 ** passes) * 1e-4 * max(1.0, abs(T)))
    res.update(status=status, obj=best, x=bestx, nodes=nodes, pivots=dic.pivots, passes=passes,
               refactors=dic.refactors)
    return res
