"""Thin object wrappers over the C ABI handles of libmldgpu (mld_model_t / mld_problem_t)."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import MAT_NAMES, EVO_NAMES, MldGpuError, check

_MAT_SHAPES = dict(A=("nx", "nx"), B1=("nx", "nu"), B2=("nx", "ndelta"), B3=("nx", "nz"), B4=("nx", "nomega"),
                   b5=("nx", 1), C=("ny", "nx"), D1=("ny", "nu"), D2=("ny", "ndelta"), D3=("ny", "nz"),
                   D4=("ny", "nomega"), d5=("ny", 1), E=("nc", "nx"), F1=("nc", "nu"), F2=("nc", "ndelta"),
                   F3=("nc", "nz"), F4=("nc", "nomega"), f5=("nc", 1), G=("nc", "ny"), Psi=("nc", "nmu"))


def mat_shape(name, dims):
    r, c = _MAT_SHAPES[name]
    return (dims[r] if isinstance(r, str) else r, dims[c] if isinstance(c, str) else c)


def evo_shapes(dims, N):
    nx, ny, nc, nw = dims["nx"], dims["ny"], dims["nc"], dims["nomega"]
    nv = dims["nu"] + dims["ndelta"] + dims["nz"] + dims["nmu"]
    return dict(Phi_x=(N * nx, nx), Gamma_v=(N * nx, N * nv), Gamma_omega=(N * nx, N * nw), Gamma_5=(N * nx, 1),
                L_x=(N * ny, nx), L_v=(N * ny, N * nv), L_omega=(N * ny, N * nw), L_5=(N * ny, 1),
                H_x=(N * nc, nx), H_v=(N * nc, N * nv), H_omega=(N * nc, N * nw), H_5=(N * nc, 1))


class GpuModel(object):
    """n_models same-shaped numeric MLD systems resident in HBM (mld_model_create)."""

    def __init__(self, mats_list, dims, time_varying=False):
        """mats_list: list of dicts name -> 2-D array (missing / empty = zeros); dims: dict with
        nx,nu,ndelta,nz,nmu,nomega,ny,nc,nu_l,nmu_l.
        time_varying: mats_list is a list of horizons, each a list of N_tilde step models (mld_model_create_tv)."""
        if isinstance(mats_list, dict):
            mats_list = [mats_list]
        self.dims = {k: int(dims.get(k, 0)) for k in ("nx", "nu", "ndelta", "nz", "nmu", "nomega", "ny", "nc", "nu_l", "nmu_l")}
        self.tv_N = 0
        n_sets = len(mats_list)
        if time_varying:
            self.tv_N = len(mats_list[0])
            if self.tv_N < 1 or any(len(h) != self.tv_N for h in mats_list):
                raise ValueError("every horizon needs the same number (>= 1) of step models")
            mats_list = [m for h in mats_list for m in h]
        self.n_models = len(mats_list)
        self.nv = self.dims["nu"] + self.dims["ndelta"] + self.dims["nz"] + self.dims["nmu"]
        self._keep = []
        ptrs = (C.POINTER(C.c_double) * 20)()
        for k, name in enumerate(MAT_NAMES):
            shp = mat_shape(name, self.dims)
            if shp[0] * shp[1] == 0:
                ptrs[k] = None
                continue
            stack = np.zeros((self.n_models,) + shp)
            any_given = False
            for i, mats in enumerate(mats_list):
                a = mats.get(name)
                if a is None or np.size(a) == 0:
                    continue
                a = np.asarray(a, dtype=np.float64)
                if a.shape != shp:
                    a = a.reshape(shp)
                stack[i] = a
                any_given = True
            if any_given:
                self._keep.append(stack)
                ptrs[k] = _lib.dptr(stack)
            else:
                ptrs[k] = None
        d = _lib.Dims(**self.dims)
        self._h = C.c_void_p()
        if self.tv_N:
            check(_lib.load().mld_model_create_tv(C.byref(self._h), C.byref(d), n_sets, self.tv_N, ptrs))
        else:
            check(_lib.load().mld_model_create(C.byref(self._h), C.byref(d), self.n_models, ptrs))
        self.n_models = n_sets
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().mld_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def condense_device(self, N_tilde, f32=False):
        """run K1+K2 into device-resident buffers (fp32 materialisation with f32=True); returns kernel milliseconds (HIP events)"""
        ms = C.c_double()
        fn = _lib.load().mld_condense_device_f32 if f32 else _lib.load().mld_condense_device
        check(fn(self._h, int(N_tilde), 0, C.byref(ms)))
        return ms.value

    def condense(self, N_tilde, names=EVO_NAMES, dtype=np.float64):
        """materialised evolution matrices, dict name -> (n_models, rows, cols) arrays; dtype np.float32 = mld_condense_f32"""
        shapes = evo_shapes(self.dims, int(N_tilde))
        f32 = np.dtype(dtype) == np.float32
        out, ptrs = {}, []
        for nm in EVO_NAMES:
            if nm in names:
                out[nm] = np.zeros((self.n_models,) + shapes[nm], dtype=np.float32 if f32 else np.float64)
                ptrs.append(out[nm].ctypes.data_as(C.POINTER(C.c_float)) if f32 else _lib.dptr(out[nm]))
            else:
                ptrs.append(None)
        check((_lib.load().mld_condense_f32 if f32 else _lib.load().mld_condense)(self._h, int(N_tilde), 0, *ptrs))
        return out


def expand_open_nodes(fix, depth, var, val, flag, pos):
    """The open nodes a stopped depth-first search leaves behind, from its stack (mld_download_open_nodes): `fix` are the fixings the search
    itself ran under (uint8 per binary, 255 = free), level k of the stack branched on decision-vector entry var[k] (pos maps it to its binary
    position), currently at val[k]; flag[k] = 1 when both children of that level are accounted for.  Returned: one fixing array per open node --
    for every level whose sibling has not been visited {levels above at their values, this level flipped}, and last the current path itself.
    The nodes are pairwise disjoint and together cover exactly what the search had not closed (tests/test_host.py)."""
    res, path = [], np.array(fix, dtype=np.uint8, copy=True)
    for k in range(int(depth)):
        kp = int(pos[int(var[k])])
        if not flag[k]:
            f = path.copy()
            f[kp] = 1 - int(val[k])
            res.append(f)
        path[kp] = int(val[k])
    res.append(path)
    return res


def make_opts(**kw):
    o = _lib.Opts()
    check(_lib.load().mld_opts_default(C.byref(o)))
    alias = dict(MIPGap="gap_rel", NodeLimit="max_nodes", IterationLimit="max_pivots", TimeLimit="time_limit")
    for k, v in kw.items():
        k = alias.get(k, k)
        if not hasattr(o, k):
            raise TypeError("unknown solver option %r" % k)
        setattr(o, k, type(getattr(o, k))(v))
    return o


class GpuProblem(object):
    """mld_problem_t: condensed constraint maps of a GpuModel + cost + solver workspace, on device."""

    def __init__(self, model, N_p, N_tilde, cost=None, **opts):
        self.model = model
        self.N_p, self.N_tilde = int(N_p), int(N_tilde)
        d = model.dims
        self.n = self.N_tilde * model.nv
        self.m = self.N_tilde * d["nc"]
        self.nW = self.N_tilde * d["nomega"]
        step_bin = np.zeros(model.nv, dtype=bool)
        step_bin[d["nu"] - d["nu_l"]:d["nu"]] = True
        step_bin[d["nu"]:d["nu"] + d["ndelta"]] = True
        omu = d["nu"] + d["ndelta"] + d["nz"]
        step_bin[omu + d["nmu"] - d["nmu_l"]:omu + d["nmu"]] = True
        self.is_bin = np.tile(step_bin, self.N_tilde)
        self.n_bin = int(self.is_bin.sum())
        self.opts = make_opts(**opts)
        self._h = C.c_void_p()
        c, keep = self._cost_struct(cost)
        check(_lib.load().mld_problem_create(C.byref(self._h), model._h, self.N_p, self.N_tilde,
                                             C.byref(c) if c is not None else None, C.byref(self.opts)))
        check(_lib.load().mld_problem_get_opts(self._h, C.byref(self.opts)))      # the size-scaled defaults, resolved
        self.batch = 0

    def _cost_struct(self, cost):
        if not cost:
            return None, []
        M, d, N = self.model.n_models, self.model.dims, self.N_tilde
        lens = dict(lin_v=(self.n,), lin_x=(N * d["nx"],), lin_y=(N * d["ny"],), quad_v=(self.n, self.n),
                    quad_x=(N * d["nx"], N * d["nx"]), quad_y=(N * d["ny"], N * d["ny"]))
        c = _lib.Cost()
        keep = []
        for k, shp in lens.items():
            a = cost.get(k)
            if a is None or int(np.prod(shp)) == 0:
                setattr(c, k, None)
                continue
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float64).reshape((-1,) + shp), (M,) + shp))
            keep.append(a)
            setattr(c, k, _lib.dptr(a))
        self._keep_cost = keep
        self._last_cost = dict(quad=tuple(cost.get(k) for k in ("quad_v", "quad_x", "quad_y")))
        return c, keep

    def set_cost(self, cost):
        c, keep = self._cost_struct(cost)
        check(_lib.load().mld_problem_set_cost(self._h, C.byref(c) if c is not None else None))

    def cost_assemble(self):
        M, n, nx, nW = self.model.n_models, self.n, self.model.dims["nx"], self.nW
        P, q0 = np.zeros((M, n, n)), np.zeros((M, n))
        Qx, Qw = np.zeros((M, n, nx)), np.zeros((M, n, nW))
        check(_lib.load().mld_cost_assemble(self._h, _lib.dptr(P), _lib.dptr(q0), _lib.dptr(Qx) if nx else None,
                                            _lib.dptr(Qw) if nW else None))
        return dict(P=P, q0=q0, Qx=Qx, Qw=Qw)

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().mld_problem_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- batch ---------------------------------------------------------------------------------
    def upload(self, x0, omega, model_idx=None, fixed_bin=None):
        d = self.model.dims
        x0 = _lib.as_f64(x0).reshape(-1, d["nx"]) if d["nx"] else np.zeros((np.shape(omega)[0] if omega is not None else 1, 0))
        batch = x0.shape[0] if d["nx"] else int(np.asarray(omega).reshape(-1, max(self.nW, 1)).shape[0])
        omega = _lib.as_f64(omega).reshape(batch, self.nW) if self.nW else np.zeros((batch, 0))
        mi = np.ascontiguousarray(model_idx, dtype=np.int32) if model_idx is not None else None
        fb = np.ascontiguousarray(fixed_bin, dtype=np.uint8).reshape(batch, self.n_bin) if fixed_bin is not None else None
        check(_lib.load().mld_upload_batch(
            self._h, batch, mi.ctypes.data_as(C.POINTER(C.c_int32)) if mi is not None else None,
            _lib.dptr(x0) if d["nx"] else None, _lib.dptr(omega) if self.nW else None,
            fb.ctypes.data_as(C.POINTER(C.c_uint8)) if fb is not None else None))
        self.batch = batch
        return batch

    def upload_constraint_blocks(self, omega_cols, col_rows=None, x_cols=None):
        """extra constraint blocks for the uploaded batch (mld_upload_constraint_blocks_x): omega_cols (batch, n_cols,
        N_tilde*nomega), col_rows (n_cols) leading rows each column constrains (None = all), x_cols (batch, n_cols, nx) the state
        each column was generated with (None = the instance's x0); None / empty clears"""
        if omega_cols is None or np.size(omega_cols) == 0:
            check(_lib.load().mld_upload_constraint_blocks(self._h, 0, None, None))
            return 0
        cols = _lib.as_f64(omega_cols).reshape(self.batch, -1, self.nW)
        n_cols = cols.shape[1]
        cr = None
        if col_rows is not None:
            cr = np.ascontiguousarray(col_rows, dtype=np.int32).reshape(n_cols)
        nx = self.model.dims["nx"]
        xc = _lib.as_f64(x_cols).reshape(self.batch, n_cols, nx) if (x_cols is not None and nx) else None
        check(_lib.load().mld_upload_constraint_blocks_x(self._h, n_cols, _lib.dptr(cols),
                                                         cr.ctypes.data_as(C.POINTER(C.c_int32)) if cr is not None else None,
                                                         _lib.dptr(xc) if xc is not None else None))
        return n_cols

    def set_opts(self, **opts):
        """limits / tolerances of the existing problem (mld_problem_set_opts): MIPGap, NodeLimit, IterationLimit, gap_abs, cut
        rounds, reserved -- no rebuild, like the per-call solver kwargs of the reference's solve()"""
        alias = dict(MIPGap="gap_rel", NodeLimit="max_nodes", IterationLimit="max_pivots", TimeLimit="time_limit")
        new = _lib.Opts.from_buffer_copy(self.opts)      # a refused call leaves the Python copy as it was, like the device options
        for k, v in opts.items():
            k = alias.get(k, k)
            if k in ("max_cuts", "n_slots", "presolve") or not hasattr(new, k):
                raise TypeError("option %r cannot be changed on an existing problem" % k)
            if k == "flags" and int(v) != int(self.opts.flags):
                raise TypeError("flags (MLD_F32) are fixed when the problem is created")
            setattr(new, k, type(getattr(new, k))(v))
        check(_lib.load().mld_problem_set_opts(self._h, C.byref(new)))
        self.opts = new

    def advance(self):
        """receding horizon on device (mld_advance_batch2): x0 <- plant update with the step-0 slice of the last solution,
        disturbance forecast moved on by one step; the next solve_resident() is the next MPC step.  Returns the number of
        instances that were NOT advanced because their solve left no usable plan (they keep state and forecast)."""
        skipped = C.c_int32(0)
        check(_lib.load().mld_advance_batch2(self._h, C.byref(skipped)))
        return int(skipped.value)

    def set_warm_start(self, bin_start):
        """MIP start of the resident batch (mld_set_warm_start): (batch, n_bin) values of the binaries, a row starting with 255 = no
        start for that instance; None clears.  A full decision vector (batch, n) is accepted too (its binaries are rounded)."""
        if bin_start is None:
            check(_lib.load().mld_set_warm_start(self._h, None))
            return
        a = np.asarray(bin_start)
        if a.ndim >= 1 and a.reshape(self.batch, -1).shape[1] == self.n and self.n != self.n_bin:
            a = np.rint(a.reshape(self.batch, self.n)[:, self.is_bin])
        a = np.ascontiguousarray(a.reshape(self.batch, self.n_bin), dtype=np.uint8)
        check(_lib.load().mld_set_warm_start(self._h, a.ctypes.data_as(C.POINTER(C.c_uint8))))

    def warm_start_from_previous(self, shift=0):
        """MIP start from the last solution, on device (mld_warm_start_from_previous): shift 0 = the plan as it is (warm_start=True of
        the reference's backend), shift k = moved k steps on with the last step repeated (after advance())"""
        check(_lib.load().mld_warm_start_from_previous(self._h, int(shift)))

    def stage(self, x0_sets, omega_sets):
        """input sets of the uploaded batch's size resident in HBM (mld_stage_inputs): x0_sets (n_sets, batch, nx), omega_sets
        (n_sets, batch, N_tilde*nomega); select(k) makes set k current without host traffic"""
        d = self.model.dims
        n_sets = int(np.shape(omega_sets)[0] if self.nW else np.shape(x0_sets)[0])
        x0s = _lib.as_f64(x0_sets).reshape(n_sets, self.batch, d["nx"]) if d["nx"] else None
        oms = _lib.as_f64(omega_sets).reshape(n_sets, self.batch, self.nW) if self.nW else None
        check(_lib.load().mld_stage_inputs(self._h, n_sets, _lib.dptr(x0s), _lib.dptr(oms)))
        return n_sets

    def select(self, k):
        check(_lib.load().mld_select_inputs(self._h, int(k)))

    def inputs(self):
        """current (x0, omega) of the resident batch"""
        d = self.model.dims
        x0, om = np.zeros((self.batch, d["nx"])), np.zeros((self.batch, self.nW))
        check(_lib.load().mld_download_inputs(self._h, _lib.dptr(x0) if d["nx"] else None, _lib.dptr(om) if self.nW else None))
        return x0, om

    def solve_resident(self):
        st = _lib.Stats()
        check(_lib.load().mld_solve_resident(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _lib.Stats._fields_}

    def use_stream(self):
        """give this problem its own HIP stream (launch / finish of several problems then overlap on the device)"""
        check(_lib.load().mld_problem_use_stream(self._h))

    def launch(self):
        """queue K3 + K5/K6 for the resident batch and return (mld_solve_launch)"""
        check(_lib.load().mld_solve_launch(self._h))

    def finish(self):
        """wait for the launched solve; statistics as solve_resident (mld_solve_finish)"""
        st = _lib.Stats()
        check(_lib.load().mld_solve_finish(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _lib.Stats._fields_}

    def download(self):
        b = self.batch
        v, obj, lbnd = np.zeros((b, self.n)), np.zeros(b), np.zeros(b)
        status, nodes, pivots = np.zeros(b, np.int32), np.zeros(b, np.int32), np.zeros(b, np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        check(_lib.load().mld_download_results(self._h, _lib.dptr(v), _lib.dptr(obj), ip(status), _lib.dptr(lbnd),
                                               ip(nodes), ip(pivots)))
        return dict(v=v, obj=obj, status=status, lower_bound=lbnd, nodes=nodes, pivots=pivots)

    def debug_trace(self, path):
        """diagnostics: every resident workgroup of the solver writes (instance, stage, pivots, queue position) into a file-backed host buffer that
        survives a GPU fault (internal entry mld_debug_trace; None switches it off).  Read it with numpy.fromfile(path, numpy.int32).reshape(-1, 16)."""
        lib = _lib.load()
        lib.mld_debug_trace.restype = C.c_int
        lib.mld_debug_trace.argtypes = [C.c_void_p, C.c_char_p]
        check(lib.mld_debug_trace(self._h, None if path is None else str(path).encode()))

    def telemetry(self):
        """per-instance in-kernel latency (ns) and dictionary rows updated; row_bytes = bytes per row"""
        b = self.batch
        lat, rows = np.zeros(b, np.int64), np.zeros(b, np.int64)
        rb = C.c_int64()
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        check(_lib.load().mld_download_telemetry(self._h, ip(lat), ip(rows), C.byref(rb)))
        return dict(latency_ns=lat, rows_updated=rows, row_bytes=int(rb.value))

    def solve(self, x0, omega, model_idx=None, fixed_bin=None, omega_cols=None, col_rows=None, x_cols=None, warm_start=None):
        self.upload(x0, omega, model_idx, fixed_bin)
        if omega_cols is not None:
            self.upload_constraint_blocks(omega_cols, col_rows, x_cols)
        if warm_start is not None:
            self.set_warm_start(warm_start)
        stats = self.solve_resident()
        out = self.download()
        out["stats"] = stats
        return out

    # -- sub-tree hand-off ---------------------------------------------------------------------------
    def set_cutoffs(self, cutoff):
        """objective (constant term included) every instance of the resident batch has to beat (mld_set_cutoffs); inf = none, None clears"""
        if cutoff is None:
            check(_lib.load().mld_set_cutoffs(self._h, None))
            return
        c = _lib.as_f64(cutoff).reshape(self.batch)
        check(_lib.load().mld_set_cutoffs(self._h, _lib.dptr(c)))

    def record_open_nodes(self, enable=True):
        check(_lib.load().mld_record_open_nodes(self._h, 1 if enable else 0))

    def open_nodes(self):
        """search stacks of the instances that stopped at a limit inside a complete search (mld_download_open_nodes):
        depth (batch; -1 = none), var / val / flag (batch, n_bin)"""
        b, nb = self.batch, max(1, self.n_bin)
        depth = np.zeros(b, np.int32)
        var, val, flag = np.zeros((b, nb), np.int16), np.zeros((b, nb), np.uint8), np.zeros((b, nb), np.uint8)
        check(_lib.load().mld_download_open_nodes(self._h, depth.ctypes.data_as(C.POINTER(C.c_int32)), var.ctypes.data_as(C.POINTER(C.c_int16)),
                                                  val.ctypes.data_as(C.POINTER(C.c_uint8)), flag.ctypes.data_as(C.POINTER(C.c_uint8))))
        return depth, var, val, flag

    def set_std_block(self, enable=True):
        """the standard constraint block in or out of the problem (mld_set_std_block): out = only the uploaded constraint blocks constrain"""
        check(_lib.load().mld_set_std_block(self._h, 1 if enable else 0))

    def set_handoff(self, enable=True, sub_nodes=0, max_gen=8, max_children=64, max_tree=160, room_factor=0.0, donate=0, rounds=0):
        """in-kernel sub-tree hand-off (mld_set_handoff): searches that stop at their node limit publish their open nodes as entries of the same
        launch's work queue; takes effect with the next upload().  sub_nodes 0 = the problem's max_nodes for items too."""
        lib = _lib.load()
        lib.mld_set_handoff.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        check(lib.mld_set_handoff(self._h, 1 if enable else 0, int(sub_nodes), int(max_gen), int(max_children), int(max_tree), float(room_factor)))
        check(lib.mld_set_handoff_policy(self._h, int(donate), int(rounds)))
        self.batch = 0 if not enable else self.batch

    def handoff_stats(self):
        out = (C.c_int64 * 4)()
        check(_lib.load().mld_handoff_stats(self._h, out))
        return dict(items=int(out[0]), given_up=int(out[1]), unfinished=int(out[2]), queue_full=int(out[3]))

    def solve_handoff_device(self, x0, omega, model_idx=None, fixed_bin=None, first_nodes=None, sub_nodes=None, max_gen=8, max_children=64, max_tree=160, room_factor=0.0, donate=0, rounds=0):
        """the batch with the hand-off inside ONE launch (set_handoff): upload, solve, download -- the merged results per instance plus `handoff`
        statistics.  The problem's own limits and the hand-off switch are restored afterwards."""
        keep_nodes = int(self.opts.max_nodes)
        try:
            if first_nodes is not None:
                self.set_opts(max_nodes=int(first_nodes))
            self.set_handoff(True, sub_nodes=int(sub_nodes or 0), max_gen=max_gen, max_children=max_children, max_tree=max_tree, room_factor=room_factor, donate=donate, rounds=rounds)
            self.upload(x0, omega, model_idx, fixed_bin)
            stats = self.solve_resident()
            out = self.download()
            out["stats"] = stats
            out["handoff"] = self.handoff_stats()
            return out
        finally:
            self.set_handoff(False)
            self.set_opts(max_nodes=keep_nodes)

    def solve_handoff(self, x0, omega, model_idx=None, fixed_bin=None, rounds=3, first_nodes=None, sub_nodes=None, max_sub=None, sub_opts=None,
                      max_open=128):
        """The batch solved with sub-tree hand-off: a first pass over all instances (node limit `first_nodes`, default the problem's), then up to
        `rounds` passes in which the OPEN NODES of the instances that stopped at the limit -- read off their depth-first stacks -- are solved as
        instances of their own (node limit `sub_nodes` each, the parent's incumbent value as cutoff), so the whole device works on the few large
        trees instead of one workgroup per tree.  An instance is proven once every one of its nodes has been closed; one whose open nodes
        outnumber `max_open` after a pass is given up (NODE_LIMIT with its incumbent and bound).  Returns the dict of
        download() (v, obj, status, lower_bound; nodes / pivots summed over all passes) plus `handoff` statistics; the resident batch afterwards
        is the last pass's sub-batch (upload again before advance() / warm starts)."""
        d = self.model.dims
        if getattr(self, "_keep_cost", None) and any(k is not None for k in (self._last_cost or {}).get("quad", ())):
            raise MldGpuError("solve_handoff: not with a quadratic cost (a stopped search records its stack only under a linear cost)")
        x0 = _lib.as_f64(x0).reshape(-1, d["nx"]) if d["nx"] else np.zeros((np.shape(omega)[0], 0))
        B = x0.shape[0] if d["nx"] else int(np.asarray(omega).reshape(-1, max(self.nW, 1)).shape[0])
        omega = _lib.as_f64(omega).reshape(B, self.nW) if self.nW else np.zeros((B, 0))
        mi = np.ascontiguousarray(model_idx, dtype=np.int32) if model_idx is not None else None
        nb = self.n_bin
        base = np.full((B, nb), 255, np.uint8) if fixed_bin is None else np.ascontiguousarray(fixed_bin, dtype=np.uint8).reshape(B, nb).copy()
        pos = np.full(self.n, -1, np.int64)
        pos[np.flatnonzero(self.is_bin)] = np.arange(nb)
        keep_nodes = int(self.opts.max_nodes)
        gap_rel, gap_abs = float(self.opts.gap_rel), float(self.opts.gap_abs)
        self.record_open_nodes(True)
        try:
            if first_nodes is not None:
                self.set_opts(max_nodes=int(first_nodes))
            out = self.solve(x0, omega, mi, None if fixed_bin is None else base)
            depth, var, val, flag = self.open_nodes()
            obj, v, status, lb = out["obj"].copy(), out["v"].copy(), out["status"].copy(), out["lower_bound"].copy()
            nodes, pivots = out["nodes"].astype(np.int64), out["pivots"].astype(np.int64)
            stats = dict(first_pass_ms=out["stats"]["solve_ms"], rounds=[], handed_off=0)

            def expand(fix, dep, vr, vl, fl):
                return expand_open_nodes(fix, dep, vr, vl, fl, pos)

            open_list = {}      # parent -> list of (fixings, lower bound valid for that node)
            for i in np.flatnonzero(status == 2):
                # a search that stopped before it became a plain depth-first search below an incumbent (deepening passes, dive, RINS) has no
                # stack that describes what is left: its one open node is the root, searched again under the incumbent's value as cutoff
                open_list[int(i)] = [(f, lb[i]) for f in (expand(base[i], depth[i], var[i], val[i], flag[i]) if depth[i] >= 0 else [base[i].copy()])]
            stats["handed_off"] = len(open_list)
            if sub_nodes is not None:
                self.set_opts(max_nodes=int(sub_nodes))
            keep_sub = {}
            if sub_opts:                                            # e.g. a shallower root cut loop for the open nodes
                keep_sub = {k: getattr(self.opts, k) for k in sub_opts}
                self.set_opts(**sub_opts)
            stuck_before = set()
            for r in range(int(rounds)):
                if not open_list:
                    break
                par = np.array([i for i, lst in open_list.items() for _ in lst], dtype=np.int64)
                fix = np.stack([f for lst in open_list.values() for f, _ in lst])
                nlb = np.array([b_ for lst in open_list.values() for _, b_ in lst])
                if max_sub is not None and par.size > max_sub:
                    break
                self.upload(x0[par], omega[par], mi[par] if mi is not None else None, fix)
                self.set_cutoffs(obj[par])
                if os.environ.get("MLD_HANDOFF_DUMP"):      # diagnostics: the sub-batch about to be solved (post-mortem replay)
                    np.savez(os.environ["MLD_HANDOFF_DUMP"], x0=x0[par], omega=omega[par], midx=(mi[par] if mi is not None else np.zeros(par.size, np.int32)), fix=fix, cutoff=obj[par], round=r)
                st = self.solve_resident()
                sub = self.download()
                d2, v2, l2, f2 = self.open_nodes()
                stats["rounds"].append(dict(sub_instances=int(par.size), parents=len(open_list), ms=st["solve_ms"]))
                rstat = stats["rounds"][-1]
                new_open, stuck = {}, set()
                for s_ in range(par.size):
                    i = int(par[s_])
                    nodes[i] += sub["nodes"][s_]; pivots[i] += sub["pivots"][s_]
                    if np.isfinite(sub["obj"][s_]) and sub["obj"][s_] < obj[i]:
                        obj[i], v[i] = sub["obj"][s_], sub["v"][s_]
                    ss = int(sub["status"][s_])
                    if ss in (0, 1):
                        continue                                    # closed: optimum of the node found, or nothing better than the cutoff in it
                    nl = max(nlb[s_], sub["lower_bound"][s_]) if np.isfinite(sub["lower_bound"][s_]) else nlb[s_]
                    if ss == 2 and d2[s_] >= 0:
                        new_open.setdefault(i, []).extend((f, nl) for f in expand(fix[s_], d2[s_], v2[s_], l2[s_], f2[s_]))
                    else:
                        # a node that could not be split (numerical trouble, no complete search yet) is retried ONCE as it is; coming back unsplit a second
                        # time it stays open for good (the tree ends NODE_LIMIT) instead of being solved again and again with the same budget
                        key = (i, fix[s_].tobytes())
                        if key in stuck_before or ss == 4:
                            dropped = stats.setdefault("dropped", 0)
                            stats["dropped"] = dropped + 1
                            gave = stats.setdefault("_gave", set()); gave.add(i)
                            continue
                        stuck_before.add(key)
                        stuck.add(i)
                        new_open.setdefault(i, []).append((fix[s_], nl))
                for i in list(open_list):
                    tol = max(gap_abs, gap_rel * abs(obj[i])) if np.isfinite(obj[i]) else 0.0
                    if i in stats.get("_gave", ()) and i not in new_open:
                        continue                                    # a node of this tree was dropped unsplit: not proven (stays NODE_LIMIT)
                    if i not in new_open:                           # every node closed: proven
                        status[i] = 0 if np.isfinite(obj[i]) else 1
                        lb[i] = min(obj[i], max(lb[i], obj[i] - tol)) if np.isfinite(obj[i]) else lb[i]
                    else:
                        lb[i] = max(lb[i], min(min(b_ for _, b_ in new_open[i]), obj[i] - tol if np.isfinite(obj[i]) else np.inf))
                open_list = {i: lst for i, lst in new_open.items()}
                if max_open is not None:                            # a tree that keeps growing is given up (it stays NODE_LIMIT with its incumbent and bound)
                    gave_up = [i for i, lst in open_list.items() if len(lst) > max_open]
                    for i in gave_up:
                        del open_list[i]
                    stats["given_up"] = stats.get("given_up", 0) + len(gave_up)
                rstat["parents_left"] = len(open_list)
                if stuck and all(i in stuck for i in open_list) and r + 1 < rounds:
                    pass                                            # (stuck nodes are simply retried with the next round's budget)
            stats["unfinished"] = len(open_list) + stats.get("given_up", 0) + len(stats.pop("_gave", ()))
            return dict(v=v, obj=obj, status=status, lower_bound=lb, nodes=nodes, pivots=pivots, stats=out["stats"], handoff=stats)
        finally:
            self.record_open_nodes(False)
            self.set_opts(max_nodes=keep_nodes)
            if sub_opts and keep_sub:
                self.set_opts(**keep_sub)
            self.set_cutoffs(None)

    def rhs(self, x0, omega, model_idx=None, scenarios=1):
        """h = H_x x_k + H_omega omega + H_5 (row-min over `scenarios` omega columns), original model"""
        d = self.model.dims
        omega = _lib.as_f64(omega)
        batch = int(np.asarray(x0).reshape(-1, d["nx"]).shape[0]) if d["nx"] else int(omega.size // max(1, scenarios * self.nW))
        x0 = _lib.as_f64(x0).reshape(batch, d["nx"]) if d["nx"] else np.zeros((batch, 0))
        omega = omega.reshape(batch, scenarios, self.nW) if self.nW else np.zeros((batch, scenarios, 0))
        mi = np.ascontiguousarray(model_idx, dtype=np.int32) if model_idx is not None else None
        h = np.zeros((batch, self.m))
        check(_lib.load().mld_rhs_batch(self._h, batch, int(scenarios),
                                        mi.ctypes.data_as(C.POINTER(C.c_int32)) if mi is not None else None,
                                        _lib.dptr(x0) if d["nx"] else None, _lib.dptr(omega) if self.nW else None, _lib.dptr(h)))
        return h
