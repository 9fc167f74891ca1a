"""helpers to read tests/golden/*.npz (made by oracle/gen_golden.py from the reference itself)"""
import glob
import os

import numpy as np

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MATS = ("A", "B1", "B2", "B3", "B4", "b5", "C", "D1", "D2", "D3", "D4", "d5", "E", "F1", "F2", "F3", "F4", "f5", "G", "Psi")
EVO_NAMES = ("Phi_x", "Gamma_v", "Gamma_omega", "Gamma_5", "L_x", "L_v", "L_omega", "L_5", "H_x", "H_v", "H_omega", "H_5")


def case_files():
    return sorted(f for f in glob.glob(os.path.join(GDIR, "ref_*.npz"))
                  if not os.path.basename(f).startswith(("ref_matrix_utils", "ref_objective")))


def load_case(path):
    z = np.load(path, allow_pickle=False)
    mats = {k: z["mat_" + k] for k in MATS}
    dims = dict(zip([str(s) for s in z["dims_names"]], [int(v) for v in z["dims_values"]]))
    dims["nc"] = dims["n_constraints"]
    return z, mats, dims, int(z["N_p"]), int(z["N_tilde"])


def check_evo(z, name, M, dims, rtol=1e-11):
    """compare a computed evolution matrix with the (possibly reduced) golden one"""
    M = np.asarray(M, dtype=np.float64)
    key = "evo_" + name
    if key in z.files:
        ref = z[key]
        if ref.size == 0 and M.size == 0 and ref.shape[0] == M.shape[0] == 0:
            return   # reference quirk: an empty constraint set gives f5/H_5 of shape (0,0) (mld_model.py:915-923)
        assert M.shape == ref.shape, (name, M.shape, ref.shape)
        scale = max(1.0, float(np.abs(ref).max())) if ref.size else 1.0
        assert np.abs(M - ref).max() <= rtol * scale if ref.size else True, name
        return
    shape = tuple(int(v) for v in z[key + "__shape"])
    assert M.shape == shape, (name, M.shape, shape)
    col0, rowlast, Mr, lM = z[key + "__col0"], z[key + "__rowlast"], z[key + "__Mr"], z[key + "__lM"]
    rng = np.random.Generator(np.random.PCG64(12345))
    rv = rng.standard_normal((shape[1], 1))
    lv = rng.standard_normal((1, shape[0]))
    scale = max(1.0, float(np.abs(col0).max()), float(np.abs(rowlast).max()))
    assert np.abs(M[:, :col0.shape[1]] - col0).max() <= rtol * scale, name
    assert np.abs(M[-rowlast.shape[0]:, :] - rowlast).max() <= rtol * scale, name
    assert np.abs(M @ rv - Mr).max() <= rtol * scale * shape[1], name
    assert np.abs(lv @ M - lM).max() <= rtol * scale * shape[0], name
