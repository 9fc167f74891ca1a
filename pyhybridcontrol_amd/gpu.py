"""Thin object wrappers over the C ABI handles of libmldgpu (mld_model_t / mld_problem_t)."""
import ctypes as C

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

    def __init__(self, mats_list, dims):
        """mats_list: list of dicts name -> 2-D array (missing / empty = zeros); dims: dict with
        nx,nu,ndelta,nz,nmu,nomega,ny,nc,nu_l,nmu_l"""
        if isinstance(mats_list, dict):
            mats_list = [mats_list]
        self.dims = {k: int(dims.get(k, 0)) for k in ("nx", "nu", "ndelta", "nz", "nmu", "nomega", "ny", "nc", "nu_l", "nmu_l")}
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
        check(_lib.load().mld_model_create(C.byref(self._h), C.byref(d), self.n_models, ptrs))
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

    def condense_device(self, N_tilde):
        """run K1+K2 into device-resident buffers; returns kernel milliseconds (HIP events)"""
        ms = C.c_double()
        check(_lib.load().mld_condense_device(self._h, int(N_tilde), 0, C.byref(ms)))
        return ms.value

    def condense(self, N_tilde, names=EVO_NAMES):
        """materialised evolution matrices, dict name -> (n_models, rows, cols) arrays"""
        shapes = evo_shapes(self.dims, int(N_tilde))
        out, ptrs = {}, []
        for nm in EVO_NAMES:
            if nm in names:
                out[nm] = np.zeros((self.n_models,) + shapes[nm])
                ptrs.append(_lib.dptr(out[nm]))
            else:
                ptrs.append(None)
        check(_lib.load().mld_condense(self._h, int(N_tilde), 0, *ptrs))
        return out
