"""Generate golden vectors by RUNNING THE REFERENCE ITSELF (in the build container only).

    python oracle/gen_golden.py            # writes tests/golden/*.npz

Uses oracle/ref_harness.py to import /root/reference unmodified (wrapt / cvxpy are stubbed: see that
file) and dumps, per case, the numeric MLD matrices and the twelve `*_N_tilde` evolution matrices the
reference's `MldEvoMatrices` computes (controllers/components/mld_evolution_matrices.py), the
`MldInfo` dims / variable types (models/mld_model.py:149-168,294-345), `block_toeplitz` /
`block_diag_dense` outputs (utils/matrix_utils.py:55-81,117-161) and the tiled objective weights of
`ObjectiveAtoms` (controllers/components/objective_atoms.py).  Large matrices are stored as reduced
views (first block column, last block row, two seeded random projections) -- enough to pin a
block-Toeplitz matrix to 1e-12.  The reference's files never leave /root/reference; only these
numeric fixtures are committed.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness  # noqa: E402

EVO = dict(state_input=("Phi_x", "Gamma_v", "Gamma_omega", "Gamma_5"), output=("L_x", "L_v", "L_omega", "L_5"),
           constraint=("H_x", "H_v", "H_omega", "H_5"))
MATS = ("A", "B1", "B2", "B3", "B4", "b5", "C", "D1", "D2", "D3", "D4", "d5", "E", "F1", "F2", "F3", "F4", "f5", "G", "Psi")
FULL_LIMIT = 40000  # elements; larger matrices are stored reduced


def reduce_matrix(name, M, nrb, ncb, out):
    """store M fully if small, else first block column + last block row + projections"""
    M = np.asarray(M, dtype=np.float64)
    if M.size <= FULL_LIMIT:
        out[name] = M
        return
    rng = np.random.Generator(np.random.PCG64(12345))
    rv = rng.standard_normal((M.shape[1], 1))
    lv = rng.standard_normal((1, M.shape[0]))
    out[name + "__shape"] = np.array(M.shape)
    out[name + "__col0"] = M[:, :ncb].copy()
    out[name + "__rowlast"] = M[-nrb:, :].copy()
    out[name + "__Mr"] = M @ rv
    out[name + "__lM"] = lv @ M


def dump_case(path, ref_model, N_p, N_tilde, extra=None):
    from controllers.components.mld_evolution_matrices import MldEvoMatrices
    info = ref_model.mld_info
    out = {}
    for k in MATS:
        out["mat_" + k] = np.asarray(ref_model[k], dtype=np.float64)
    dims = {k: int(info[k]) for k in ("nx", "nu", "ndelta", "nz", "nmu", "nomega", "ny", "n_constraints", "nv",
                                       "nu_l", "ndelta_l", "nz_l", "nmu_l", "nv_l")}
    out["dims_names"] = np.array(list(dims.keys()))
    out["dims_values"] = np.array(list(dims.values()))
    out["var_type_v"] = np.array([str(t) for t in np.asarray(info["var_type_v"]).ravel()])
    out["N_p"], out["N_tilde"] = np.array(N_p), np.array(N_tilde)
    evo = MldEvoMatrices(None, N_p=N_p, N_tilde=N_tilde, mld_numeric_k=ref_model, mld_numeric_tilde=None)
    rows = dict(state_input=dims["nx"], output=dims["ny"], constraint=dims["n_constraints"])
    for typ, names in EVO.items():
        for nm in names:
            M = evo[typ][nm + "_N_tilde"]
            ncb = {"x": dims["nx"], "v": dims["nv"], "omega": dims["nomega"], "5": 1}[nm.split("_")[1]]
            reduce_matrix("evo_" + nm, M, max(rows[typ], 1), max(ncb, 1), out)
            Mp = evo[typ][nm + "_N_p"]
            out["evoNp_shape_" + nm] = np.array(np.asarray(Mp).shape)
    if extra:
        out.update(extra)
    np.savez_compressed(path, **out)
    print("wrote", os.path.relpath(path, ROOT), sum(np.asarray(v).nbytes for v in out.values()) // 1024, "KiB")


def main():
    ref_harness.install()
    import warnings
    warnings.simplefilter("ignore")
    from models.mld_model import MldModel
    from utils.matrix_utils import block_toeplitz, block_diag_dense
    from examples.residential_mg_with_pv_and_dewhs.modelling.micro_grid_models import DewhModel, GridModel, PvModel
    from controllers.components.objective_atoms import ObjectiveAtoms
    from pyhybridcontrol_amd import synthetic as syn

    gdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gdir, exist_ok=True)

    # 1. the example's own models through the reference's sympy -> numeric pipeline
    dewh = DewhModel(const_heat=True).mld_numeric
    for N in (3, 5, 25):
        dump_case(os.path.join(gdir, "ref_dewh_N%d.npz" % N), dewh, N - 1, N)
    for nd in (0, 1, 3, 7):
        g = GridModel(num_devices=nd).mld_numeric
        for N in (3, 25):
            dump_case(os.path.join(gdir, "ref_grid%d_N%d.npz" % (nd, N)), g, N - 1, N)
    dump_case(os.path.join(gdir, "ref_pv_N3.npz"), PvModel().mld_numeric, 2, 3)

    # 2. synthetic clusters (BASELINE cfg1/cfg2/cfg3 shapes) pushed through the reference's MldModel + MldEvoMatrices
    for name in ("cfg1", "cfg2", "cfg3"):
        cfg = syn.CONFIGS[name]
        rng = np.random.Generator(np.random.PCG64(cfg["seed"] * 1000))
        mats, dims, params = syn.make_agent(cfg["n_h"], rng, tie=cfg["tie"])
        ref = MldModel(dict(mats), nu_l=dims["nu_l"], ts=900)
        dump_case(os.path.join(gdir, "ref_%s.npz" % name), ref, cfg["N_p"], cfg["N_p"] + 1)

    # 3. seeded random MLDs, including zero-sized dimensions
    shapes = [dict(nx=3, nu=2, nd=1, nz=1, nmu=2, nw=2, ny=2, nc=5), dict(nx=0, nu=1, nd=2, nz=1, nmu=0, nw=1, ny=1, nc=4),
              dict(nx=2, nu=0, nd=0, nz=0, nmu=0, nw=1, ny=1, nc=0), dict(nx=4, nu=3, nd=0, nz=2, nmu=1, nw=0, ny=0, nc=6),
              dict(nx=1, nu=1, nd=1, nz=1, nmu=1, nw=1, ny=1, nc=3), dict(nx=5, nu=2, nd=3, nz=2, nmu=4, nw=3, ny=2, nc=9),
              dict(nx=2, nu=2, nd=0, nz=0, nmu=2, nw=0, ny=2, nc=4), dict(nx=3, nu=0, nd=2, nz=2, nmu=0, nw=2, ny=0, nc=5),
              dict(nx=6, nu=4, nd=2, nz=1, nmu=3, nw=2, ny=3, nc=8), dict(nx=2, nu=1, nd=1, nz=0, nmu=0, nw=1, ny=1, nc=2)]
    for idx, sh in enumerate(shapes):
        rng = np.random.Generator(np.random.PCG64(7000 + idx))

        def rnd(r, c, dens=0.7):
            return rng.standard_normal((r, c)) * (rng.random((r, c)) < dens)
        m = {}
        if sh["nx"]:
            m["A"] = 0.4 * rnd(sh["nx"], sh["nx"], 1.0)
            m["b5"] = rnd(sh["nx"], 1)
        for nm, r, c in (("B1", "nx", "nu"), ("B2", "nx", "nd"), ("B3", "nx", "nz"), ("B4", "nx", "nw"),
                         ("D1", "ny", "nu"), ("D2", "ny", "nd"), ("D3", "ny", "nz"), ("D4", "ny", "nw"),
                         ("E", "nc", "nx"), ("F1", "nc", "nu"), ("F2", "nc", "nd"), ("F3", "nc", "nz"),
                         ("F4", "nc", "nw"), ("G", "nc", "ny"), ("Psi", "nc", "nmu"), ("C", "ny", "nx")):
            if sh[r] and sh[c]:
                m[nm] = rnd(sh[r], sh[c])
        if sh["ny"]:
            m["d5"] = rnd(sh["ny"], 1)
            if "C" not in m:
                m["C"] = np.zeros((sh["ny"], sh["nx"]))
        if sh["nc"]:
            m["f5"] = rng.standard_normal((sh["nc"], 1)) + 3.0
        nu_l = int(rng.integers(0, sh["nu"] + 1)) if sh["nu"] else 0
        ref = MldModel(dict(m), nu_l=nu_l, ts=1)
        N = int(rng.integers(2, 7))
        dump_case(os.path.join(gdir, "ref_rand%02d.npz" % idx), ref, N - 1, N)

    # 4. block_toeplitz / block_diag_dense raw behaviour
    rng = np.random.Generator(np.random.PCG64(99))
    blocks = [rng.standard_normal((2, 3)) for _ in range(4)]
    zero = [np.zeros((2, 3))] * 4
    out = dict(blocks=np.array(blocks), toeplitz=block_toeplitz(blocks, zero), diag_small=block_diag_dense([blocks[0]] * 3),
               diag_big=block_diag_dense([rng.standard_normal((9, 11))] * 8)[:18, :22])
    np.savez_compressed(os.path.join(gdir, "ref_matrix_utils.npz"), **out)

    # 5. objective-atom weights (parsing + tiling) on the DEWH model (nx=1,nu=1,nmu=2) and cfg2 cluster
    N_p, N_t = 4, 5
    specs = {
        "a": {"q_mu": [10, 1], "q_u": 1.0, "Q_x": [[2.0]], "Q_x_f": [[5.0]]},
        "b": {"q_u": np.arange(1, N_t + 1, dtype=float), "q_Quadratic_x": 3.0, "Q_mu": [[1.0, 0.2], [0.2, 2.0]]},
        "c": {"q_u_N_p": [1.0, 2.0, 3.0, 4.0], "q_mu_f": [7.0, 8.0], "q_x": 0.0},
        "d": {"Q_Linear_mu": [[1.0, 2.0], [3.0, 4.0]], "q_Quadratic_mu": [2.0, 3.0]},
    }
    wout = {}
    for tag, spec in specs.items():
        oa = ObjectiveAtoms(None, N_p=N_p, N_tilde=N_t, mld_numeric_k=dewh, objective_atoms_struct=dict(spec))
        for var, atoms in oa.items():
            if atoms:
                for an, a in atoms.items():
                    wout["%s|%s|%s" % (tag, var, an)] = np.asarray(a.weight.weight_N_tilde, dtype=np.float64)
        wout["spec_" + tag] = np.array(sorted(spec.keys()))
        for k, v in spec.items():
            wout["specval_%s|%s" % (tag, k)] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(gdir, "ref_objective_weights.npz"), **wout)
    print("wrote objective weights:", [k for k in wout if not k.startswith("spec")])
    ref_harness.uninstall()


if __name__ == "__main__":
    main()
