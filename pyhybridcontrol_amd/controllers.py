"""Host-side mirror of the reference's MPC controller API, backed by libmldgpu (no cvxpy, no CPU solver).

Mirrors, for the hot path only:
  * ``MldEvoMatrices``            controllers/components/mld_evolution_matrices.py:19-250
  * ``ConstraintSolvedController`` / ``MpcController``
                                  controllers/controller_base.py:149-548, controllers/mpc_controller.py:19-101
  * ``ControllerBuildRequiredError`` / ``ControllerSolverError``   controller_base.py:25-30
with the same method names, argument meaning and error behaviour:
``build()`` -> problem on the GPU; ``solve()`` -> objective value (float) or ControllerSolverError;
``feedback()`` -> variables at step k; ``sim_step_k()`` -> plant step + log; ``external_solve``
bypasses the backend (controller_base.py:507,536-538).  Where the reference builds cvxpy expressions,
this keeps numeric arrays.  Scenario / min-max constraint blocks, disable_soft_constraints, L1 / Linf and rate atoms run on
the same kernels (row-min right-hand sides, model augmentation); unsupported reference features raise
NotImplementedError loudly (rate atoms on mu, time-varying horizons, custom standard constraints).
"""
import time

import numpy as np

from . import gpu
from ._lib import MldGpuError, STATUS_NAMES
from .mld_model import MldModel
from .mld_model import ParNotSet
from .objective_atoms import ObjectiveAtoms, atleast_2d_col, matmul


class ControllerBuildRequiredError(RuntimeError):
    pass


class ControllerSolverError(RuntimeError):
    pass


_EVO = dict(state_input=("Phi_x", "Gamma_v", "Gamma_omega", "Gamma_5"), output=("L_x", "L_v", "L_omega", "L_5"),
            constraint=("H_x", "H_v", "H_omega", "H_5"))
_ROWS = dict(state_input="nx", output="ny", constraint="nc")


def check_numeric_tilde(mld_numeric_tilde, N_tilde):
    """the list of N_tilde numeric step models of a time-varying horizon (mld_numeric_tilde, controller_base.py:311;
    every step must have the dimensions and variable types of step 0)"""
    tilde = [m if isinstance(m, MldModel) else MldModel(m) for m in mld_numeric_tilde]
    if len(tilde) != N_tilde:
        raise ValueError("mld_numeric_tilde needs one model per horizon step: %d given, N_tilde = %d" % (len(tilde), N_tilde))
    d0 = tilde[0].mld_info.as_gpu_dims()
    for k, m in enumerate(tilde):
        if m.mld_info.as_gpu_dims() != d0:
            raise ValueError("mld_numeric_tilde[%d] differs from step 0 in its dimensions or variable types" % k)
    return tilde


class MldEvoMatrices(dict):
    """{'state_input'|'output'|'constraint': {name_N_tilde, name_N_p}} computed by kernels K1+K2.

    x_N_tilde = Phi_x x(0) + Gamma_v v_N_tilde + Gamma_omega omega_N_tilde + Gamma_5
    y_N_tilde = L_x x(0) + L_v v_N_tilde + L_omega omega_N_tilde + L_5
    H_v v_N_tilde <= H_x x(0) + H_omega omega_N_tilde + H_5          (mld_evolution_matrices.py:59-67)
    """

    def __init__(self, controller=None, N_p=None, N_tilde=None, mld_numeric_k=None, mld_numeric_tilde=None):
        super(MldEvoMatrices, self).__init__()
        self._controller = controller
        self.N_p = N_p if N_p is not None else controller.N_p
        self.N_tilde = N_tilde if N_tilde is not None else (controller.N_tilde if controller else self.N_p + 1)
        if mld_numeric_tilde is None and controller is not None and mld_numeric_k is None:
            mld_numeric_tilde = controller.mld_numeric_tilde
        # one model per horizon step (component_base.py:72-81: step 0 then plays the part of mld_numeric_k)
        self._tilde = check_numeric_tilde(mld_numeric_tilde, self.N_tilde) if mld_numeric_tilde else None
        self._model = self._tilde[0] if self._tilde else (mld_numeric_k if mld_numeric_k is not None else controller.mld_numeric_k)
        if not isinstance(self._model, MldModel):
            raise ValueError("mld_numeric_k must be an MldModel with mld_type=='numeric'")
        self._gpu_model = None
        self.update(reset=True)

    @property
    def mld_info_k(self):
        return self._model.mld_info

    @property
    def state_input(self):
        return self["state_input"]

    @property
    def output(self):
        return self["output"]

    @property
    def constraint(self):
        return self["constraint"]

    def gpu_model(self):
        if self._gpu_model is None:
            dims = self._model.mld_info.as_gpu_dims()
            if self._tilde:
                self._gpu_model = gpu.GpuModel([[m.as_mats() for m in self._tilde]], dims, time_varying=True)
            else:
                self._gpu_model = gpu.GpuModel([self._model.as_mats()], dims)
        return self._gpu_model

    def update(self, reset=False, **_):
        if not reset and self:
            return
        dims = self._model.mld_info.as_gpu_dims()
        evo = self.gpu_model().condense(self.N_tilde)
        for typ, names in _EVO.items():
            rows = dims[_ROWS[typ]]
            blk = {}
            for nm in names:
                M = evo[nm][0]
                blk[nm + "_N_tilde"] = M
                blk[nm + "_N_p"] = M[:self.N_p * rows, :]           # row-prefix views (:246-250)
            self[typ] = blk

    def get_evo_matrices_N_tilde(self, N_tilde=None):
        """row-prefix slices for a shorter constraint horizon (:89-105)"""
        if N_tilde is None or N_tilde == self.N_tilde:
            return self
        if N_tilde > self.N_tilde:
            raise ValueError("N_tilde:%d cannot be greater than self.N_tilde:%d" % (N_tilde, self.N_tilde))
        out = dict.__new__(MldEvoMatrices)
        dict.__init__(out)
        out.__dict__.update(self.__dict__)
        dims = self._model.mld_info.as_gpu_dims()
        for typ in _EVO:
            rows = dims[_ROWS[typ]]
            out[typ] = {k: (v[:N_tilde * rows, :] if k.endswith("N_tilde") else v) for k, v in self[typ].items()}
        return out


class EvoConstraint(object):
    """numeric stand-in for the cvxpy constraint  LHS @ v <= RHS  returned by gen_evo_constraints"""

    def __init__(self, H_v, rhs, omega_cols=None, rows=None, x_is_parameter=True, x_problem=None):
        self.H_v, self.rhs = H_v, rhs
        # what the block was generated from: disturbance columns (N_tilde*nomega, S) padded to the full horizon, the
        # number of constraint rows it spans, and whether x_k was the controller's parameter (updated at solve time) or an
        # explicit state (controller_base.py:411-416), kept in the GPU problem's state layout
        self.omega_cols, self.rows, self.x_is_parameter = omega_cols, rows, x_is_parameter
        self.x_problem = x_problem


class MldSimLog(dict):
    """k -> dict of column vectors (controller_base.py:58-146), without the DataFrame sugar"""

    def set_sim_k(self, k, sim_k=None, **kwargs):
        self.pop(k, None)
        self.update_sim_k(k, sim_k, **kwargs)

    def update_sim_k(self, k, sim_k=None, **kwargs):
        if sim_k is not None and not isinstance(sim_k, dict):
            raise TypeError("sim_k must be subtype of dict or None, not: %r" % type(sim_k).__name__)
        entry = dict(self.get(k, {}))
        for name, val in dict(sim_k or {}, **kwargs).items():
            if val is not None:
                val = atleast_2d_col(val)
                if name in entry and entry[name].shape != val.shape:
                    raise ValueError("shape of var_k must match previous inserts")
                entry[name] = val
        dict.__setitem__(self, k, entry)

    def __setitem__(self, k, sim_k):
        self.set_sim_k(k=k, sim_k=sim_k)

    def get_concat_log(self, add_column_levels=None):
        """one pandas DataFrame over all logged steps: index k, columns (var_names, var_index); steps that never logged
        a variable show NaN there (controller_base.py:116-146)"""
        import pandas as pd
        index = sorted(self)
        shapes = {}
        for k in index:
            for name, val in self[k].items():
                shapes.setdefault(name, val.shape[0])
        frames = {}
        for name, dim in shapes.items():
            if dim == 0:
                continue
            numeric = all(np.issubdtype(self[k][name].dtype, np.number) or self[k][name].dtype == bool
                          for k in index if name in self[k])
            block = np.full((len(index), dim), np.nan if numeric else None, dtype=float if numeric else object)
            for r, k in enumerate(index):
                if name in self[k]:
                    block[r] = self[k][name][:, 0]
            frames[name] = pd.DataFrame(block)
        df = pd.concat(frames, keys=list(frames), axis=1)
        df.columns.names = ["var_names", "var_index"]
        df.index = index
        df.index.name = "k"
        if add_column_levels:
            df = pd.concat([df], keys=[add_column_levels], axis=1)
        return df


class MpcController(object):
    def __init__(self, model=None, x_k=None, omega_tilde_k=None, N_p=None, N_tilde=None, agent=None, mld_numeric=None,
                 mld_numeric_tilde=None, **solver_opts):
        if agent is not None:
            raise NotImplementedError("agent ownership layer (models/agents.py) is out of scope; pass model=")
        self._N_p = N_p if N_p is not None else 0
        self._N_tilde = N_tilde if N_tilde is not None else self._N_p + 1
        # time-varying horizon: one numeric model per step; the reference keeps the attribute (controller_base.py:175,
        # 306-312) but never sets it -- here it is a constructor argument and a settable property
        self._tilde = check_numeric_tilde(mld_numeric_tilde, self._N_tilde) if mld_numeric_tilde else None
        model = model if model is not None else mld_numeric
        if model is None and self._tilde:
            model = self._tilde[0]
        if not isinstance(model, MldModel):
            model = MldModel(model)
        self._sim_model = model                    # the plant sim_step_k evolves (controller_base.py:235)
        self._model = self._tilde[0] if self._tilde else model
        self._solver_opts = dict(solver_opts)
        # handoff=dict(first_nodes=..., sub_nodes=..., [max_gen, max_children, max_tree]): the in-kernel sub-tree hand-off (mld_set_handoff) -- the
        # one instance a solve() call holds spreads over the idle workgroups of the device once its search has run first_nodes nodes (the reference's
        # backend uses every core of its machine on that one tree, controller_base.py:509).  None / absent = one workgroup per solve.
        self._handoff = self._solver_opts.pop("handoff", None)
        self._sim_log = MldSimLog()
        self._solve_time_overall = 0
        self._solve_time_solver = 0
        self.reset_components(x_k=x_k, omega_tilde_k=omega_tilde_k)

    # -- state ----------------------------------------------------------------------------------
    def reset_components(self, x_k=None, omega_tilde_k=None):
        info = self.mld_info_k
        self._build_required = True
        self._mld_evo_matrices = None
        self._std_obj_atoms = ObjectiveAtoms(info.as_gpu_dims(), self._N_p, self._N_tilde)
        self._problem = None
        self._epi_blocks, self._epi_sig, self._epi_dims, self._epi_model, self._vmap = [], (), None, None, None
        self._rate_vars, self._rate_info, self._rate_dims, self._k_neg1 = [], {}, None, {}
        self._sense = 1.0
        self._std_custom, self._other_objectives = None, []
        self._solution = None
        self._solution_problem, self._k_solved = None, None      # last solution in the GPU problem's layout (the MIP start of the next solve)
        self._x_k = np.zeros((info.nx, 1))
        self._omega_tilde_k = np.zeros((info.nomega * self._N_tilde, 1))
        if x_k is not None:
            self.x_k = x_k
        if omega_tilde_k is not None:
            self.omega_tilde_k = omega_tilde_k

    @property
    def N_p(self):
        return self._N_p

    @property
    def N_tilde(self):
        return self._N_tilde

    @property
    def mld_numeric_k(self):
        return self._model

    @property
    def mld_numeric_tilde(self):
        return self._tilde

    @mld_numeric_tilde.setter
    def mld_numeric_tilde(self, value):
        """a new horizon of step models (e.g. shifted by one step between MPC iterations): same shapes, so the cost atoms,
        the logs and the state are kept; the condensed maps and the GPU problem are rebuilt at the next build()"""
        tilde = check_numeric_tilde(value, self._N_tilde) if value else None
        new0 = tilde[0] if tilde else self._sim_model
        if new0.mld_info.as_gpu_dims() != self._model.mld_info.as_gpu_dims():
            raise ValueError("mld_numeric_tilde must keep the controller's dimensions and variable types")
        self._tilde, self._model = tilde, new0
        self._mld_evo_matrices = None
        if self._problem is not None:
            self._problem.close(); self._problem = None
        self._solution_problem = None
        if getattr(self, "_epi_model", None) is not None:
            self._epi_model.close(); self._epi_model = None
        self._build_required = True

    def _step_mats(self):
        """per-step matrices of a time-varying horizon, or None"""
        return [m.as_mats() for m in self._tilde] if self._tilde else None

    @property
    def mld_info_k(self):
        return self._model.mld_info

    @property
    def mld_evo_matrices(self):
        if self._mld_evo_matrices is None:
            self._mld_evo_matrices = MldEvoMatrices(self)
        return self._mld_evo_matrices

    @property
    def std_obj_atoms(self):
        return self._std_obj_atoms

    @property
    def sim_log(self):
        return self._sim_log

    @property
    def build_required(self):
        return self._build_required

    def set_build_required(self):
        self._build_required = True

    @property
    def x_k(self):
        return self._x_k

    @x_k.setter
    def x_k(self, value):
        self._x_k = self._check_shape("x_k", value, (self.mld_info_k.nx, 1))

    @property
    def omega_tilde_k(self):
        return self._omega_tilde_k

    @omega_tilde_k.setter
    def omega_tilde_k(self, value):
        self._omega_tilde_k = self._check_shape("omega_tilde_k", value, (self.mld_info_k.nomega * self._N_tilde, 1))

    @staticmethod
    def _check_shape(name, value, required):
        value = atleast_2d_col(value)
        if value.dtype == np.object_:
            raise TypeError("'new_value' must be a numeric array like object or None.")
        if 0 in required:
            return np.empty(required)
        if value.shape != required:
            raise ValueError("Incorrect shape:%s for %s, a shape of %s is required." % (value.shape, name, required))
        return value.astype(np.float64)

    # -- objective / constraints -------------------------------------------------------------------
    def set_std_obj_atoms(self, objective_atoms_struct=None, **kwargs):
        self._std_obj_atoms.set(objective_atoms_struct, **kwargs)
        self._build_required = True

    def update_std_obj_atoms(self, objective_weights_struct=None, **kwargs):
        self._std_obj_atoms.update(objective_weights_struct, **kwargs)
        self._build_required = True

    def gen_evo_constraints(self, x_k=None, omega_tilde_k=None, omega_scenarios_k=None, N_tilde=None):
        """H_v v <= H_x x_k + H_omega omega + H_5 ; with omega_scenarios_k the row-min over scenario
        columns of H_omega @ Omega (controller_base.py:411-456).  RHS computed by kernel K3."""
        x_is_parameter = x_k is None
        x_k = self._x_k if x_k is None else self._check_shape("x_k", x_k, (self.mld_info_k.nx, 1))
        N_t = self._N_tilde if N_tilde is None else N_tilde
        if not N_t <= self._N_tilde:
            raise ValueError("N_tilde: %d must be less or equal to self.N_tilde: %d" % (N_t, self._N_tilde))
        info = self.mld_info_k
        if not info.n_constraints:
            return EvoConstraint(np.zeros((0, self._N_tilde * info.nv)), np.zeros((0, 1)))
        prob = self._ensure_problem()
        if omega_scenarios_k is not None:
            Om = np.asarray(omega_scenarios_k, dtype=np.float64)
            if Om.shape[0] == N_t * info.nomega and N_t < self._N_tilde:
                Om = np.vstack([Om, np.zeros(((self._N_tilde - N_t) * info.nomega, Om.shape[1]))])
            h = prob.rhs(self._x_problem(x_k).T, Om.T[np.newaxis], scenarios=Om.shape[1])[0]
            cols = Om
        else:
            om = self._omega_tilde_k if omega_tilde_k is None else atleast_2d_col(omega_tilde_k)
            if om.shape[0] == N_t * info.nomega and N_t < self._N_tilde:
                om = np.vstack([om, np.zeros(((self._N_tilde - N_t) * info.nomega, 1))])
            h = prob.rhs(self._x_problem(x_k).T, om.T)[0]
            cols = om
        rows = N_t * info.n_constraints
        H_v = self.mld_evo_matrices.constraint["H_v_N_tilde"][:rows]
        h = h[self._orig_rows(self._N_tilde * info.n_constraints)]
        return EvoConstraint(H_v, h[:rows].reshape(-1, 1), omega_cols=np.array(cols, dtype=np.float64), rows=rows,
                             x_is_parameter=x_is_parameter,
                             x_problem=None if x_is_parameter else np.array(self._x_problem(x_k), dtype=np.float64).reshape(-1))

    def set_constraints(self, std_evo_constaints=ParNotSet, other_constraints=ParNotSet, disable_soft_constraints=False):
        """controller_base.py:457-475.  `other_constraints`: blocks from gen_evo_constraints (scenario columns, min / max
        disturbance profiles, reduced N_tilde); they share the standard block's left-hand side, so on the GPU they
        become extra right-hand-side columns reduced by a row-wise minimum (mld_upload_constraint_blocks)."""
        if std_evo_constaints is not ParNotSet:
            # None = the standard block (controller_base.py:460-462); a list = those blocks INSTEAD of it -- [] leaves only `other_constraints`
            # (build(with_std_constraints=False), mpc_controller.py:84-87).  On the GPU: mld_set_std_block(0) + the blocks as right-hand-side columns.
            if std_evo_constaints is None:
                self._std_custom = None
            else:
                blocks = list(std_evo_constaints)
                for b in blocks:
                    if not isinstance(b, EvoConstraint) or b.omega_cols is None:
                        raise TypeError("std_evo_constaints must come from gen_evo_constraints()")
                self._std_custom = blocks
        if other_constraints is not ParNotSet:
            blocks = list(other_constraints or [])
            for b in blocks:
                if not isinstance(b, EvoConstraint) or b.omega_cols is None:
                    raise TypeError("other_constraints must come from gen_evo_constraints()")
            self._other_constraints = blocks
        # mu == 0 (controller_base.py:466-471): extra model rows mu <= 0, see epigraph.hard_block
        self._no_soft = bool(disable_soft_constraints and self.mld_info_k.nmu)
        self._build_required = True

    def _epi_signature(self, blocks):
        return tuple((b["var"], b["M"].shape, b["S"].shape, b["M"].tobytes(), bool(b.get("one_sided"))) for b in blocks)

    def _augmentation(self):
        """(mats', dims', blocks, rate variables, rate info): the MLD model the GPU problem is built from.  Rate atoms add
        lag states and rate outputs (epigraph.augment_rates), L1 / Linf atoms and disable_soft_constraints add rows and
        auxiliaries (epigraph.augment); without any of them it is the controller's own model."""
        from . import epigraph
        W = self._std_obj_atoms.weights
        dims0, N = self._model.mld_info.as_gpu_dims(), self._N_tilde
        rv = epigraph.rate_vars(W)
        steps = self._step_mats()
        if steps:                              # time-varying horizon: the same augmentation of every step model
            aug = [epigraph.augment_rates(m, dims0, rv) if rv else (m, dims0, {}) for m in steps]
            mats1, dims1, info = [a[0] for a in aug], aug[0][1], aug[0][2]
        else:
            mats1, dims1, info = epigraph.augment_rates(self._model.as_mats(), dims0, rv) if rv else (self._model.as_mats(), dims0, {})
        blocks = epigraph.plan(W, dims0, N)
        for b in blocks:                      # atoms on x / y act on the original entries, not on lag states / rate outputs
            if b["var"] in ("x", "y"):
                width = dims1["nx"] if b["var"] == "x" else dims1["ny"]
                b["M"] = np.hstack([b["M"], np.zeros((b["M"].shape[0], width - b["M"].shape[1]))])
        return mats1, dims1, blocks, rv, info

    def _problem_cost(self, cost):
        """the cost dict in the layouts of the GPU problem (lag states / rate outputs / epigraph auxiliaries appended)"""
        from . import epigraph
        if not (self._epi_blocks or self._rate_vars):
            return cost
        dims0, N = self._model.mld_info.as_gpu_dims(), self._N_tilde
        c = epigraph.lift_xy_cost(cost or {}, dims0, self._rate_dims, N)
        if self._rate_vars:
            epigraph.rate_cost_and_blocks(self._std_obj_atoms.weights, self._rate_dims, self._rate_info, N, c)
        lifted, self._vmap = epigraph.lift_cost(c, self._rate_dims, self._epi_dims, N, self._epi_blocks)
        if not self._epi_blocks:
            self._vmap = None
        return lifted

    def _ensure_problem(self):
        from . import epigraph
        mats1, dims1, blocks, rv, info = self._augmentation()
        N = self._N_tilde
        if rv:
            blocks = blocks + epigraph.rate_cost_and_blocks(self._std_obj_atoms.weights, dims1, info, N,
                                                            dict(lin_y=np.zeros(N * dims1["ny"]), quad_y=None))
        if getattr(self, "_no_soft", False):
            blocks = blocks + [epigraph.hard_block(dims1, N)]
        sig = (tuple(rv), self._epi_signature(blocks))
        if self._problem is not None and sig != getattr(self, "_epi_sig", ()):
            self._problem.close()                    # the set of augmenting atoms changed: another augmented model
            self._problem = None
            self._solution_problem = None
            if getattr(self, "_epi_model", None) is not None:
                self._epi_model.close()
                self._epi_model = None
        if self._problem is None:
            evo = self.mld_evo_matrices
            cost = self._std_obj_atoms.to_cost()
            self._omega_atoms = cost.pop("_omega_atoms", [])
            self._epi_blocks, self._epi_sig, self._vmap = blocks, sig, None
            self._rate_vars, self._rate_info, self._rate_dims = rv, info, dims1
            if blocks or rv:
                if isinstance(mats1, list):
                    aug = [epigraph.augment(m, dims1, blocks) for m in mats1]
                    self._epi_dims = aug[0][1]
                    self._epi_model = gpu.GpuModel([[a[0] for a in aug]], self._epi_dims, time_varying=True)
                else:
                    mats2, self._epi_dims, _ = epigraph.augment(mats1, dims1, blocks)
                    self._epi_model = gpu.GpuModel([mats2], self._epi_dims)
                self._problem = gpu.GpuProblem(self._epi_model, self._N_p, self._N_tilde, self._signed(self._problem_cost(cost)), **self._solver_opts)
            else:
                self._epi_dims = dims1
                self._problem = gpu.GpuProblem(evo.gpu_model(), self._N_p, self._N_tilde, self._signed(cost), **self._solver_opts)
            if self._handoff:
                ho = dict(self._handoff)
                ho.pop("first_nodes", None)
                self._problem.set_handoff(True, **ho)
        return self._problem

    def _x_problem(self, x_k=None):
        """the state handed to the GPU problem: x_k followed by the lag states' initial values (variables_k_neg1)"""
        x = self._x_k if x_k is None else x_k
        if not getattr(self, "_rate_vars", None):
            return x
        parts = [x]
        for v in self._rate_vars:
            k = self._rate_info[v][2]
            val = self._k_neg1.get(v)
            parts.append(np.zeros((k, 1)) if val is None else atleast_2d_col(val).reshape(k, 1))
        return np.vstack(parts)

    @property
    def variables_k_neg1(self):
        """values at step k-1 that rate atoms ('d<var>') difference against (variables.py:88-102); missing -> zeros"""
        return dict(self._k_neg1)

    @variables_k_neg1.setter
    def variables_k_neg1(self, struct):
        self._k_neg1 = {k: atleast_2d_col(v) for k, v in dict(struct or {}).items() if v is not None}

    def _orig_rows(self, n_rows):
        """indices of the original constraint rows among the problem's rows (epigraph rows follow them in every step)"""
        nc = self.mld_info_k.n_constraints
        if not getattr(self, "_epi_blocks", None):
            return np.arange(n_rows)
        nc2 = self._epi_dims["nc"]
        return np.concatenate([np.arange(k * nc2, k * nc2 + nc) for k in range(n_rows // nc)])

    def _signed(self, cost):
        return {k: (None if v is None else self._sense * np.asarray(v)) for k, v in cost.items()}

    def set_objective(self, std_objective=ParNotSet, other_objectives=ParNotSet):
        """mpc_controller.py:63-74.  The reference adds cvxpy expressions; here `other_objectives` is a list of further atom sets -- ObjectiveAtoms
        objects or dicts in the string-keyed form of set_std_obj_atoms -- whose costs are ADDED to the standard objective (linear and quadratic atoms;
        L1 / Linf / rate atoms belong in the standard set, which carries the model augmentation they need)."""
        if std_objective is not ParNotSet and std_objective is not None and std_objective != 0:
            raise NotImplementedError("a custom std_objective expression: set atoms with set_std_obj_atoms() instead")
        if other_objectives is not ParNotSet:
            extra = []
            for oa in (other_objectives or []):
                if not isinstance(oa, ObjectiveAtoms):
                    atoms = ObjectiveAtoms(self.mld_info_k.as_gpu_dims(), self._N_p, self._N_tilde)
                    atoms.set(oa)
                    oa = atoms
                if oa.epigraph_blocks() or any(k[3] for k in oa.weights):
                    raise NotImplementedError("L1 / Linf / rate atoms in other_objectives")
                extra.append(oa)
            self._other_objectives = extra
        self._build_required = True

    @staticmethod
    def _add_costs(a, b):
        out = dict(a)
        for k, v in b.items():
            if k == "_omega_atoms":
                out[k] = list(out.get(k, [])) + list(v)
            elif v is not None:
                out[k] = v if out.get(k) is None else out[k] + v
        return out

    def build(self, with_std_objective=True, with_std_constraints=True, sense=None, disable_soft_constraints=False):
        """mpc_controller.py:76-101"""
        if not with_std_constraints:
            self.set_constraints(std_evo_constaints=[], disable_soft_constraints=disable_soft_constraints)
        else:
            self.set_constraints(std_evo_constaints=None, disable_soft_constraints=disable_soft_constraints)
        sense = "minimize" if sense is None else sense
        if sense.lower().startswith("min"):
            self._sense = 1.0
        elif sense.lower().startswith("max"):
            self._sense = -1.0
        else:
            raise ValueError("Problem 'sense' must be either 'minimize' or 'maximize', got '%s'." % sense)
        cost = self._std_obj_atoms.to_cost() if with_std_objective else {}
        for oa in getattr(self, "_other_objectives", []):
            cost = self._add_costs(cost, oa.to_cost())
        self._omega_atoms = cost.pop("_omega_atoms", []) if cost else []
        self._ensure_problem()
        self._problem.set_std_block(getattr(self, "_std_custom", None) is None)
        has_norms = any(not b.get("one_sided") for b in self._epi_blocks) or bool(self._rate_vars)
        if has_norms and not with_std_objective:
            raise NotImplementedError("with_std_objective=False while L1 / Linf atoms are set")
        if has_norms and self._sense < 0:
            raise ValueError("L1 / Linf atoms are convex: the problem cannot be maximised")
        cost = self._problem_cost(cost) if (cost or self._epi_blocks or self._rate_vars) else cost
        self._problem.set_cost(self._signed(cost) if cost else None)
        self._build_required = False

    # -- solve -------------------------------------------------------------------------------------
    def solve(self, k, x_k=None, omega_tilde_k=None, external_solve=None, solver=None, verbose=False, warm_start=True,
              parallel=False, *args, method=None, **kwargs):
        """controller_base.py:491-540.  `solver`, `parallel`, `method` are accepted for signature compatibility and ignored
        (there is one backend); the Gurobi-style kwargs MIPGap / NodeLimit / IterationLimit / TimeLimit the reference forwards
        (micro_grid_control_simulation.py:232) are honoured per call, TimeLimit as seconds of device time per instance.
        `warm_start` (default True, forwarded by the reference to its backend, :493,509-512): the previous solution's binaries are the
        MIP start of this solve, moved on by k - k_previous steps when both are integers (the receding-horizon shift; 0 for a
        re-solve of the same step)."""
        start = time.time()
        try:
            if x_k is not None:
                self.x_k = x_k
            if omega_tilde_k is not None:
                self.omega_tilde_k = omega_tilde_k
            if self._build_required:
                raise ControllerBuildRequiredError(
                    "%s problem has not been built or needs to be rebuilt." % self.__class__.__name__)
            if external_solve is not None:
                self._solve_time_solver = 0
                return external_solve
            remap = {}
            for key, val in kwargs.items():
                if key in ("MIPGap", "NodeLimit", "IterationLimit", "TimeLimit"):
                    remap[key] = val
                else:
                    raise TypeError("unsupported solver option %r" % key)
            # Gurobi-style kwargs are PER CALL, as in the reference (they go straight to Problem.solve, controller_base.py:509): the
            # limits of the existing problem are set without a rebuild and fall back to the constructor's options afterwards
            names = dict(MIPGap="gap_rel", NodeLimit="max_nodes", IterationLimit="max_pivots", TimeLimit="time_limit")
            ctor = gpu.make_opts(**self._solver_opts)
            eff = {k2: getattr(ctor, k2) for k2 in ("gap_rel", "max_nodes", "max_pivots", "gap_abs", "time_limit")}
            if self._handoff and self._handoff.get("first_nodes"):
                eff["max_nodes"] = int(self._handoff["first_nodes"])      # (with the hand-off the limit is per queue entry: the instance's own search, then sub_nodes per open node)
            eff.update({names[k2]: v2 for k2, v2 in remap.items()})
            cur = self._problem.opts
            if any(getattr(cur, k2) != type(getattr(cur, k2))(v2) for k2, v2 in eff.items()):
                self._problem.set_opts(**eff)
            try:
                cols = rows = xcols = None
                all_blocks = list(getattr(self, "_std_custom", None) or []) + list(getattr(self, "_other_constraints", None) or [])
                if all_blocks:
                    cols = np.hstack([b.omega_cols for b in all_blocks]).T[np.newaxis]
                    nc = self.mld_info_k.n_constraints
                    nc_p = self._epi_dims["nc"] if self._epi_blocks else nc
                    rows = np.concatenate([np.full(b.omega_cols.shape[1], (b.rows // nc) * nc_p) for b in all_blocks])
                    if any(not b.x_is_parameter for b in all_blocks):
                        xp = self._x_problem().reshape(-1)       # blocks generated with an explicit x_k keep it; the others follow the parameter
                        xcols = np.vstack([np.tile((xp if b.x_is_parameter else b.x_problem), (b.omega_cols.shape[1], 1))
                                           for b in all_blocks])[np.newaxis]
                mip_start = None
                if warm_start and self._solution_problem is not None:
                    shift = 0
                    if isinstance(k, (int, np.integer)) and isinstance(self._k_solved, (int, np.integer)):
                        shift = max(0, int(k) - int(self._k_solved))
                    V = self._solution_problem.reshape(self._N_tilde, -1)
                    if shift:
                        V = np.vstack([V[min(shift, self._N_tilde - 1):], np.repeat(V[-1:], min(shift, self._N_tilde - 1), axis=0)])
                    mip_start = V.reshape(1, -1)
                out = self._problem.solve(self._x_problem().T, self._omega_tilde_k.T, omega_cols=cols, col_rows=rows, x_cols=xcols,
                                          warm_start=mip_start)
            except MldGpuError as e:
                self._solve_time_solver = np.nan
                raise ControllerSolverError(str(e)) from e
            self._solve_time_solver = out["stats"]["solve_ms"] * 1e-3
            status = STATUS_NAMES[int(out["status"][0])]
            solution = self._sense * float(out["obj"][0])
            if verbose:
                print("mldgpu: status=%s objective=%r nodes=%d pivots=%d" % (status, solution, out["nodes"][0], out["pivots"][0]))
            if not np.isfinite(solution):
                raise ControllerSolverError("solve() failed with objective: '%s', and status: %s" % (solution, status))
            v = out["v"][0]
            self._solution_problem, self._k_solved = np.array(v, dtype=np.float64), k
            self._solution = (v if self._vmap is None else v[self._vmap]).reshape(-1, 1)
            self._status = status
            return solution
        finally:
            self._solve_time_overall = time.time() - start

    def feedback(self, k, x_k=None, omega_tilde_k=None, external_solve=None, **kwargs):
        self.solve(k=k, x_k=x_k, omega_tilde_k=omega_tilde_k, external_solve=external_solve, **kwargs)
        return self.variables_k

    @property
    def variables_k(self):
        """step-k slices {x,u,delta,z,mu,v,y,omega} as column vectors (variables.py:75-85)"""
        info = self.mld_info_k
        out = dict(x=self._x_k, omega=self._omega_tilde_k[:info.nomega])
        if self._solution is None:
            return out
        v = self._solution[:info.nv]
        o1, o2, o3 = info.nu, info.nu + info.ndelta, info.nu + info.ndelta + info.nz
        out.update(v=v, u=v[:o1], delta=v[o1:o2], z=v[o2:o3], mu=v[o3:])
        m = self._model
        out["y"] = (matmul(m["C"], out["x"]) + matmul(m["D1"], out["u"]) + matmul(m["D2"], out["delta"]) + matmul(m["D3"], out["z"])
                    + matmul(m["D4"], out["omega"]) + m["d5"])
        return out

    @property
    def v_N_tilde(self):
        return self._solution

    def predicted_trajectory(self):
        """x_N_tilde and y_N_tilde of the last solution (variables.py:259-275) from the K1/K2 matrices"""
        evo = self.mld_evo_matrices
        s, o = evo.state_input, evo.output
        x = matmul(s["Phi_x_N_tilde"], self._x_k) + matmul(s["Gamma_v_N_tilde"], self._solution) + \
            matmul(s["Gamma_omega_N_tilde"], self._omega_tilde_k) + s["Gamma_5_N_tilde"]
        y = matmul(o["L_x_N_tilde"], self._x_k) + matmul(o["L_v_N_tilde"], self._solution) + \
            matmul(o["L_omega_N_tilde"], self._omega_tilde_k) + o["L_5_N_tilde"]
        return x, y

    def sim_step_k(self, k, x_k=None, u_k=None, omega_k=None, mld_numeric_k=None, solver=None, step_state=True):
        """controller_base.py:229-253.  When the plant is the control model driven by the MPC's own (x, u, omega),
        the solve's delta/z/mu are the auxiliaries and are reused; any override (another plant model, another x, u
        or omega) resolves them from (x, u, omega) like the reference does at every step -- on the GPU
        (MldModel._compute_aux -> aux_resolve.AuxResolver)."""
        var_k = self.variables_k
        reuse = mld_numeric_k is None and x_k is None and u_k is None and omega_k is None
        omega_k = omega_k if omega_k is not None else var_k["omega"]
        x_k = x_k if x_k is not None else var_k["x"]
        u_k = u_k if u_k is not None else var_k.get("u")
        sim_model = mld_numeric_k if mld_numeric_k is not None else self._model
        if reuse:
            lsim_k = sim_model.lsim_k(x_k=x_k, u_k=u_k, delta_k=var_k.get("delta"), z_k=var_k.get("z"),
                                      mu_k=var_k.get("mu"), omega_k=omega_k)
        else:
            lsim_k = sim_model.lsim_k(x_k=x_k, u_k=u_k, omega_k=omega_k, solver=solver)
        if step_state:
            lsim_k.update({name + "_hat": val for name, val in var_k.items()})
            self._sim_log.set_sim_k(k=k, sim_k=lsim_k)
            self._sim_log.update_sim_k(k=k, time_solve_overall=self._solve_time_overall,
                                       time_in_solver=self._solve_time_solver)
            self.x_k = lsim_k["x_k1"]
        else:
            del lsim_k["x_k1"]
        return lsim_k
