"""Numeric MLD model container: host-side mirror of the reference's ``MldInfo`` / ``MldModel``
(models/mld_model.py:108-388, 390-995), numeric half only.

    x(k+1) = A x + B1 u + B2 delta + B3 z + B4 omega + b5
    y(k)   = C x + D1 u + D2 delta + D3 z + D4 omega + d5
    E x + F1 u + F2 delta + F3 z + F4 omega + G y + Psi mu <= f5 ,   mu >= 0        (:456-463)

Same constructor keywords, the same dimension / variable-type rules, the same zero / empty padding
and shape errors.  The symbolic / callable pipeline (sympy, CallableMatrix) and StructDict sugar are
out of scope: matrices are plain read-only ndarrays.
"""
import numpy as np


from .objective_atoms import atleast_2d_col, matmul


class _ParNotSetType(object):
    """sentinel of the reference's utils.helper_funcs.ParNotSet: "argument not supplied" (None means zeros)"""
    _inst = None

    def __new__(cls):
        if cls._inst is None:
            cls._inst = object.__new__(cls)
        return cls._inst

    def __repr__(self):
        return "ParNotSet"

    def __bool__(self):
        return False


ParNotSet = _ParNotSetType()

STATE_INPUT_MATS = ("A", "B1", "B2", "B3", "B4", "b5")
OUTPUT_MATS = ("C", "D1", "D2", "D3", "D4", "d5")
CONSTRAINT_MATS = ("E", "F1", "F2", "F3", "F4", "f5", "G", "Psi")
SYS_MAT_NAMES = STATE_INPUT_MATS + OUTPUT_MATS + CONSTRAINT_MATS

# (row system dim, column var dim) of every matrix; var dim None = offset vector
_MAT_DIMS = dict(A=("n_states", "nx"), B1=("n_states", "nu"), B2=("n_states", "ndelta"), B3=("n_states", "nz"),
                 B4=("n_states", "nomega"), b5=("n_states", None),
                 C=("n_outputs", "nx"), D1=("n_outputs", "nu"), D2=("n_outputs", "ndelta"), D3=("n_outputs", "nz"),
                 D4=("n_outputs", "nomega"), d5=("n_outputs", None),
                 E=("n_constraints", "nx"), F1=("n_constraints", "nu"), F2=("n_constraints", "ndelta"),
                 F3=("n_constraints", "nz"), F4=("n_constraints", "nomega"), f5=("n_constraints", None),
                 G=("n_constraints", "ny"), Psi=("n_constraints", "nmu"))
_MAT_TYPE = {**{k: "state_input" for k in STATE_INPUT_MATS}, **{k: "output" for k in OUTPUT_MATS},
             **{k: "constraint" for k in CONSTRAINT_MATS}}


class MldInfo(dict):
    """dims, binary dims and variable types (models/mld_model.py:149-168, 294-345)"""
    _var_names = ("x", "u", "delta", "z", "omega", "y", "mu", "v")
    _controllable_var_names = ("u", "delta", "z", "mu")

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def get_var_dim(self, var_name):
        return self["n" + var_name]

    def get_var_bin_dim(self, var_name):
        return self["n" + var_name + "_l"]

    def get_var_type(self, var_name):
        return self["var_type_" + var_name]

    def as_gpu_dims(self):
        return dict(nx=self["nx"], nu=self["nu"], ndelta=self["ndelta"], nz=self["nz"], nmu=self["nmu"],
                    nomega=self["nomega"], ny=self["ny"], nc=self["n_constraints"], nu_l=self["nu_l"], nmu_l=self["nmu_l"])


class MldModel(object):
    def __init__(self, system_matrices=None, ts=None, param_struct=None, bin_dims_struct=None, var_types_struct=None,
                 **kwargs):
        bin_dims = dict(bin_dims_struct or {})
        for k in list(kwargs):
            if k in ("nu_l", "nmu_l", "nx_l", "nomega_l", "ny_l"):
                bin_dims[k] = kwargs.pop(k)
        if system_matrices and kwargs:
            raise ValueError("Individual matrix arguments cannot be set if 'system_matrices' argument is set")
        creation = dict(system_matrices or kwargs)
        for name in creation:
            if name not in SYS_MAT_NAMES:
                raise ValueError("Invalid matrix name in system_matrices: {}".format(name))
        mats, shapes = {}, {}
        for name in SYS_MAT_NAMES:
            m = creation.get(name)
            if m is None:
                mats[name] = np.empty((0, 0))
            else:
                m = atleast_2d_col(m)
                if not np.issubdtype(m.dtype, np.number):
                    raise TypeError("System matrices must be numeric, callable, or symbolic.")
                mats[name] = m
            shapes[name] = mats[name].shape
        if creation.get("C") is None:                     # C defaults to I (mld_model.py:515-520)
            mats["C"] = np.eye(*shapes["A"])
            shapes["C"] = mats["C"].shape
        dims = self._get_mld_dims(shapes)
        self._validate_and_pad(mats, shapes, dims)
        for m in mats.values():
            m.setflags(write=False)
        self._mats = mats
        self._all_zero_mats = {k for k, m in mats.items() if np.all(m == 0)}
        self._all_empty_mats = {k for k, m in mats.items() if m.size == 0}
        self._mld_info = self._make_info(dims, bin_dims, var_types_struct or {}, ts, param_struct)
        self._version = 0

    # -- container ------------------------------------------------------------------------------
    def __getitem__(self, k):
        return self._mats[k]

    def __getattr__(self, k):
        mats = self.__dict__.get("_mats", {})
        if k in mats:
            return mats[k]
        raise AttributeError(k)

    def keys(self):
        return self._mats.keys()

    def items(self):
        return self._mats.items()

    @property
    def mld_info(self):
        return self._mld_info

    @property
    def mld_type(self):
        return "numeric"

    def to_numeric(self, *a, **k):
        return self

    def as_mats(self):
        """dict name -> float64 2-D array (for the GPU upload)"""
        return {k: np.asarray(v, dtype=np.float64) for k, v in self._mats.items()}

    # -- dims (mld_model.py:149-168) ------------------------------------------------------------
    @staticmethod
    def _get_mld_dims(shapes):
        def rows(names):
            return max(shapes[n][0] for n in names)

        def cols(names):
            return max(shapes[n][1] for n in names)
        d = dict(n_states=rows(STATE_INPUT_MATS), n_outputs=rows(OUTPUT_MATS), n_constraints=rows(CONSTRAINT_MATS),
                 nx=rows(STATE_INPUT_MATS), ny=rows(OUTPUT_MATS),
                 nu=cols(("B1", "D1", "F1")), ndelta=cols(("B2", "D2", "F2")), nz=cols(("B3", "D3", "F3")),
                 nomega=cols(("B4", "D4", "F4")), nmu=cols(("Psi",)))
        # the reference takes the max of the three hstack widths *before* padding (mld_model.py:163-166), which
        # under-counts when e.g. F1 is omitted; nv is defined here as the width v = [u;delta;z;mu] really has
        d["nv"] = d["nu"] + d["ndelta"] + d["nz"] + d["nmu"]
        return d

    @staticmethod
    def _validate_and_pad(mats, shapes, dims):
        """shape rules and zero/empty padding of models/mld_model.py:869-952"""
        A_shape, C_shape = shapes["A"], shapes["C"]
        if A_shape[0] != A_shape[1] and 0 not in A_shape:
            raise ValueError("Invalid shape for state matrix A:'{}', must be a square matrix or scalar".format(A_shape))
        for name in SYS_MAT_NAMES:
            sys_dim_name, var_dim_name = _MAT_DIMS[name]
            sys_dim = dims[sys_dim_name]
            var_dim = dims.get(var_dim_name) if var_dim_name else None
            shp = shapes[name]
            mtype = _MAT_TYPE[name]
            if 0 not in shp:
                if mtype == "state_input" and shp[0] != A_shape[0] and 0 not in A_shape:
                    raise ValueError("Invalid shape for state_input matrix/vector '{}':{}, must have same row "
                                     "dimension as state matrix 'A', i.e. '({}, *)'".format(name, shp, A_shape[0]))
                if mtype == "output" and shp[0] != C_shape[0] and 0 not in C_shape:
                    raise ValueError("Invalid shape for output matrix/vector '{}':{}, must have same row "
                                     "dimension as output matrix 'C', i.e. '({}, *)'".format(name, shp, C_shape[0]))
                if var_dim is not None and shp[1] != var_dim:
                    raise ValueError("Invalid shape for {} matrix/vector '{}':{}, column dimension must be equal to "
                                     "var dim '{}':{}, i.e. '(*, {})'".format(mtype, name, shp, var_dim_name, var_dim, var_dim))
                if var_dim is None and shp[1] != 1:
                    raise ValueError("'{}' must be of column vector, scalar or null array, currently has shape:{}".format(name, shp))
                if shp[0] != sys_dim:
                    raise ValueError("Invalid shape for {} matrix/vector '{}':{}, row dimension must be equal to system "
                                     "dimension - '{}':{}, i.e. '({}, *)'.".format(mtype, name, shp, sys_dim_name, sys_dim, sys_dim))
            else:
                vd = var_dim
                if vd is None and mtype != "constraint":
                    vd = 1
                new_shape = (sys_dim, vd) if vd and vd > 0 else (sys_dim, 0)
                mats[name] = np.zeros(new_shape)
                shapes[name] = new_shape
        if dims["n_constraints"] and 0 in shapes["f5"]:
            raise ValueError("Constraint vector 'f5' can only be null if all constraint matrices are null.")

    @staticmethod
    def _make_info(dims, bin_dims, var_types, ts, param_struct):
        """binaries are the trailing n*_l entries of u / mu, every delta, no z (mld_model.py:294-345)"""
        info = MldInfo(dims)
        info["ts"] = ts
        info["param_struct"] = param_struct
        for var in MldInfo._var_names:
            if var == "v":
                continue
            dim = info["n" + var]
            vt = var_types.get("var_type_" + var)
            if var == "delta":
                nb = dim
            elif var == "z":
                nb = 0
            else:
                nb = bin_dims.get("n" + var + "_l")
                if vt is not None:
                    vt = atleast_2d_col(np.asarray(vt, dtype=str))
                    if vt.size != dim:
                        raise ValueError("Dimension of 'var_type_{0}' must match dimension: 'n{0}'".format(var))
                    if not set(vt.ravel()) <= {"c", "b"}:
                        raise ValueError("All elements of var_type_vectors must be in {'c', 'b'}")
                    nvt = int((vt == "b").sum())
                    if nb is not None and nb != nvt:
                        raise ValueError("Number of binary variables in var_type_vect does not match dimension of 'n%s_l'" % var)
                    nb = nvt
                nb = int(nb or 0)
            if nb > dim:
                raise ValueError("Value of 'n{0}_l':{1} must be non-negative value <= dimension 'n{0}':{2}".format(var, nb, dim))
            if var in ("u", "mu") and vt is not None:
                # the GPU path (and the reference's own 'c'*(dim-bin)+'b'*bin rule, :331) keeps binaries trailing
                expect = np.array(list("c" * (dim - nb) + "b" * nb)).reshape(-1, 1)
                if not np.array_equal(vt, expect):
                    raise NotImplementedError("binary entries of '%s' must be the trailing ones" % var)
            info["n" + var + "_l"] = nb
            info["var_type_" + var] = np.array(list("c" * (dim - nb) + "b" * nb)).reshape(-1, 1)
        info["nv_l"] = sum(info["n" + v + "_l"] for v in MldInfo._controllable_var_names)
        info["var_type_v"] = np.vstack([info["var_type_" + v] for v in MldInfo._controllable_var_names]) \
            if info["nv"] else np.empty((0, 1), dtype=str)
        return info

    # -- one-step simulation with known auxiliaries (models/mld_model.py:647-699) ---------------------
    def lsim_k(self, x_k=ParNotSet, u_k=ParNotSet, delta_k=ParNotSet, z_k=ParNotSet, mu_k=ParNotSet, v_k=ParNotSet,
               omega_k=ParNotSet, solver=None, cons_tol=1e-6):
        """x(k+1), y(k) and hard-constraint satisfaction for given (x, u, omega) (models/mld_model.py:647-699).
        ``None`` means zeros; auxiliaries left at ``ParNotSet`` are resolved like the reference's ``_compute_aux``
        (:701-766) -- here as a horizon-1 instance of the GPU path (aux_resolve.AuxResolver; no CPU fallback)."""
        info = self._mld_info
        if omega_k is ParNotSet and info.nomega:
            raise ValueError("variable omega_k cannot be set to ParNotSet")
        if v_k is not ParNotSet and not all(a is ParNotSet for a in (u_k, delta_k, z_k, mu_k)):
            raise ValueError("Either supply concatenated input in 'v_k' or supply individual inputs "
                             "'u_k', 'delta_k', 'z_k' and 'mu_k', but not both.")
        if v_k is ParNotSet:
            if u_k is ParNotSet and info.nu:
                raise ValueError("variable u_k cannot be set to ParNotSet")
            unknown = tuple(n for n, a in (("delta", delta_k), ("z", z_k), ("mu", mu_k))
                            if a is ParNotSet and info["n" + n] > 0)
            if unknown:
                delta_k, z_k, mu_k = self._compute_aux(x_k=x_k, u_k=u_k, delta_k=delta_k, z_k=z_k, mu_k=mu_k,
                                                       omega_k=omega_k, unknown=unknown)
        x_k, u_k, delta_k, z_k, mu_k, omega_k = [None if a is ParNotSet else a for a in (x_k, u_k, delta_k, z_k, mu_k, omega_k)]
        if v_k is ParNotSet:
            v_k = None

        def col(v, dim):
            if v is None or dim == 0:
                return np.zeros((dim, 1))
            return atleast_2d_col(v).reshape(dim, 1)
        x_k = col(x_k, info.nx)
        omega_k = col(omega_k, info.nomega)
        if v_k is not None:
            v_k = atleast_2d_col(v_k)
            o1, o2, o3 = info.nu, info.nu + info.ndelta, info.nu + info.ndelta + info.nz
            u_k, delta_k, z_k, mu_k = v_k[:o1], v_k[o1:o2], v_k[o2:o3], v_k[o3:]
        u_k, delta_k, z_k, mu_k = col(u_k, info.nu), col(delta_k, info.ndelta), col(z_k, info.nz), col(mu_k, info.nmu)
        m = self._mats
        x_k1 = matmul(m["A"], x_k) + matmul(m["B1"], u_k) + matmul(m["B2"], delta_k) + matmul(m["B3"], z_k) + matmul(m["B4"], omega_k) + m["b5"]
        y_k = matmul(m["C"], x_k) + matmul(m["D1"], u_k) + matmul(m["D2"], delta_k) + matmul(m["D3"], z_k) + matmul(m["D4"], omega_k) + m["d5"]
        f5 = m["f5"] if m["f5"].size else np.zeros((info.n_constraints, 1))
        cons = (matmul(m["E"], x_k) + matmul(m["F1"], u_k) + matmul(m["F2"], delta_k) + matmul(m["F3"], z_k) + matmul(m["F4"], omega_k) + matmul(m["G"], y_k)
                + matmul(m["Psi"], (mu_k * 0)) - f5 <= cons_tol)          # hard-constraint satisfaction, as :692-694
        return dict(x_k1=x_k1, x=x_k, u=u_k, delta=delta_k, z=z_k, mu=mu_k, v=np.vstack((u_k, delta_k, z_k, mu_k)),
                    y=y_k, omega=omega_k, cons=cons)

    def _compute_aux(self, x_k=ParNotSet, u_k=ParNotSet, delta_k=ParNotSet, z_k=ParNotSet, mu_k=ParNotSet,
                     omega_k=ParNotSet, unknown=None, solver=None):
        """the (delta, z, mu) completing (x, u, omega): feasibility MIP of models/mld_model.py:701-766 on the GPU.
        Infeasible -> NaN columns, as the reference returns when the solver yields no value (:757-763)."""
        from .aux_resolve import AuxResolver
        info = self._mld_info
        if unknown is None:
            unknown = tuple(n for n, a in (("delta", delta_k), ("z", z_k), ("mu", mu_k)) if a is ParNotSet and info["n" + n] > 0)
        key = tuple(unknown)
        cache = self.__dict__.setdefault("_aux_resolvers", {})
        if key not in cache:
            cache[key] = AuxResolver(self.as_mats(), info.as_gpu_dims(), unknown=key)
        res = cache[key]

        def row(a, dim):
            if a is ParNotSet or a is None or dim == 0:
                return None if a is ParNotSet else np.zeros((1, dim))
            return atleast_2d_col(a).reshape(1, dim)
        out = res.resolve(np.zeros((1, info.nx)) if row(x_k, info.nx) is None else row(x_k, info.nx),
                          np.zeros((1, info.nu)) if row(u_k, info.nu) is None else row(u_k, info.nu),
                          np.zeros((1, info.nomega)) if row(omega_k, info.nomega) is None else row(omega_k, info.nomega),
                          delta=row(delta_k, info.ndelta), z=row(z_k, info.nz), mu=row(mu_k, info.nmu))
        return tuple(out[n].reshape(-1, 1) for n in ("delta", "z", "mu"))


def gen_schedule_params_tilde(N_tilde, param_struct, schedule_params_evo=None, **kwargs):
    """{name: [value_0 .. value_{N_tilde-1}]} -> one parameter subset per horizon step
    (PvMldSystemModel._gen_schedule_params_tilde, models/mld_model.py:1181-1208; same errors)."""
    evo = dict(schedule_params_evo) if schedule_params_evo is not None else {}
    evo.update({k: v for k, v in kwargs.items() if k in param_struct})
    if not evo:
        return None
    tilde = [dict.fromkeys(evo.keys()) for _ in range(N_tilde)]
    for name, values in evo.items():
        if name not in param_struct:
            raise ValueError("Invalid schedule_param_name:'%s' in schedule_params_evo, name needs to be present in "
                             "param_struct." % name)
        if len(values) != N_tilde:
            raise ValueError("Invalid length:'%d' for schedule_param_tilde:'%s', length of schedule_param_tilde must be "
                             "equal to N_tilde:'%d'" % (len(values), name, N_tilde))
        for k, v in enumerate(values):
            tilde[k][name] = v
    return tilde


def get_mld_numeric_tilde(get_mld_numeric, N_tilde, param_struct=None, schedule_params_tilde=None):
    """The step models of a time-varying horizon (PvMldSystemModel.get_mld_numeric_tilde, models/mld_model.py:1210-1227).
    ``get_mld_numeric(param_struct)`` is the numeric model factory -- the reference's is the lambdified symbolic model,
    which is outside this package; any callable returning an MldModel (or its matrices) does.  Without a schedule the
    one model is repeated, as in the reference."""
    base = dict(param_struct or {})

    def make(ps):
        m = get_mld_numeric(ps)
        return m if isinstance(m, MldModel) else MldModel(m)

    if schedule_params_tilde is None:
        return [make(base)] * N_tilde
    if len(schedule_params_tilde) != N_tilde:
        raise ValueError("Invalid length:'%d' for param_struct_tilde.schedule_param_tilde, length of schedule_param_tilde "
                         "must be equal to N_tilde:'%d'" % (len(schedule_params_tilde), N_tilde))
    return [make(dict(base, **schedule_params_tilde[k])) for k in range(N_tilde)]
