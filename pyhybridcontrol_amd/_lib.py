"""ctypes binding of libmldgpu.so (include/mldgpu.h).  No torch, no numpy compute: arrays in, arrays out.

The library is the product; there is no CPU fallback.  Importing this module never touches the GPU;
the first call that needs a device raises MldGpuError if none is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLDGPU_LIB") or os.path.join(_HERE, "libmldgpu.so")      # (MLDGPU_LIB: a diagnostic build of the same library, scripts/ only)

MAT_NAMES = ("A", "B1", "B2", "B3", "B4", "b5", "C", "D1", "D2", "D3", "D4", "d5",
             "E", "F1", "F2", "F3", "F4", "f5", "G", "Psi")
EVO_NAMES = ("Phi_x", "Gamma_v", "Gamma_omega", "Gamma_5", "L_x", "L_v", "L_omega", "L_5",
             "H_x", "H_v", "H_omega", "H_5")
STATUS_NAMES = {0: "optimal", 1: "infeasible", 2: "node_limit", 3: "numerical", 4: "unbounded"}
COMM_ID_BYTES = 128
MLD_F32 = 1


class MldGpuError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("nx", "nu", "ndelta", "nz", "nmu", "nomega", "ny", "nc", "nu_l", "nmu_l")]


class Opts(C.Structure):
    _fields_ = [("gap_abs", C.c_double), ("gap_rel", C.c_double), ("max_nodes", C.c_int32),
                ("max_pivots", C.c_int32), ("cut_rounds", C.c_int32), ("cuts_per_round", C.c_int32),
                ("max_cuts", C.c_int32), ("presolve", C.c_int32), ("n_slots", C.c_int32), ("mir_per_round", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32), ("time_limit", C.c_double)]


class Cost(C.Structure):
    _fields_ = [(k, C.POINTER(C.c_double)) for k in ("lin_v", "lin_x", "lin_y", "quad_v", "quad_x", "quad_y")]


class Stats(C.Structure):
    _fields_ = [("nodes", C.c_int64), ("pivots", C.c_int64), ("cuts", C.c_int64), ("refactors", C.c_int64),
                ("n_optimal", C.c_int32), ("n_infeasible", C.c_int32), ("n_node_limit", C.c_int32),
                ("n_numerical", C.c_int32), ("solve_ms", C.c_double), ("rhs_ms", C.c_double)]


_lib = None

# every symbol include/mldgpu.h declares (tests check that the shared object exports all of them)
EXPORTS = ("mld_device_count", "mld_set_device", "mld_last_error", "mld_version", "mld_device_info",
           "mld_opts_size", "mld_model_create", "mld_model_create_tv", "mld_model_destroy", "mld_condense_device", "mld_condense", "mld_condense_device_f32", "mld_condense_f32", "mld_opts_default",
           "mld_problem_create", "mld_problem_set_cost", "mld_problem_destroy", "mld_cost_assemble",
           "mld_solve_batch", "mld_upload_batch", "mld_upload_constraint_blocks", "mld_upload_constraint_blocks_x", "mld_solve_resident", "mld_problem_use_stream", "mld_solve_launch", "mld_solve_finish", "mld_download_results", "mld_download_telemetry",
           "mld_rhs_batch", "mld_problem_set_opts", "mld_problem_get_opts", "mld_advance_batch", "mld_advance_batch2", "mld_set_warm_start", "mld_warm_start_from_previous", "mld_set_cutoffs", "mld_record_open_nodes", "mld_download_open_nodes", "mld_set_handoff", "mld_handoff_stats", "mld_set_handoff_policy", "mld_set_std_block", "mld_download_inputs", "mld_stage_inputs", "mld_select_inputs", "mld_gather_results",
           "mld_comm_unique_id", "mld_comm_init", "mld_gather", "mld_comm_destroy")


def load():
    """dlopen the in-tree library (built by pyhybridcontrol_amd.build); raises if it is missing"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MldGpuError("libmldgpu.so is not built (run `python -m pyhybridcontrol_amd.build`); "
                          "there is no CPU fallback for the MPC solve path")
    lib = C.CDLL(LIB_PATH)
    lib.mld_last_error.restype = C.c_char_p
    lib.mld_version.restype = C.c_char_p
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("mld_last_error", "mld_version"):
            fn.restype = C.c_int
    if int(lib.mld_opts_size()) != C.sizeof(Opts):
        raise MldGpuError("libmldgpu.so was built with another layout of mld_opts (%d bytes, this binding %d): rebuild it (python -m pyhybridcontrol_amd.build)"
                          % (lib.mld_opts_size(), C.sizeof(Opts)))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise MldGpuError("libmldgpu error %d: %s" % (rc, load().mld_last_error().decode()))


def device_count():
    return int(load().mld_device_count())


def version():
    return load().mld_version().decode()


def device_info():
    name = C.create_string_buffer(64)
    ncu, hbm, lds = C.c_int(), C.c_int64(), C.c_int()
    check(load().mld_device_info(name, 64, C.byref(ncu), C.byref(hbm), C.byref(lds)))
    return dict(name=name.value.decode(), n_cu=ncu.value, hbm_bytes=hbm.value, lds_bytes=lds.value)


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a
