"""The oracle (numpy + C restatements) against golden vectors produced by the REFERENCE ITSELF
(oracle/gen_golden.py): condensing, MldInfo dims/types (a9), weights (a14)."""
import os

import numpy as np
import pytest

import _golden as g
import condense_np as cn
import orc


@pytest.mark.parametrize("path", g.case_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_condense_numpy_and_c_match_reference(path):
    z, mats, dims, N_p, N_t = g.load_case(path)
    d = cn.mld_dims(mats)
    for k in ("nx", "nu", "ndelta", "nz", "nmu", "nomega", "ny", "nc", "nv"):
        assert d[k] == dims[k], (k, d[k], dims[k])
    evo = cn.condense(mats, N_t)
    for name in g.EVO_NAMES:
        g.check_evo(z, name, evo[name], dims)
        # *_N_p matrices are row prefixes (mld_evolution_matrices.py:246-250)
        rows = dict(Phi=dims["nx"], Gamma=dims["nx"], L=dims["ny"], H=dims["nc"])[name.split("_")[0]]
        np_shape = tuple(int(v) for v in z["evoNp_shape_" + name])
        assert np_shape[0] == evo[name][:N_p * rows].shape[0]
        assert np_shape[1] == evo[name].shape[1] or np_shape[0] == 0
    evo_c = orc.condense(mats, dict(dims, nu_l=dims["nu_l"]), N_t)
    for name in g.EVO_NAMES:
        g.check_evo(z, name, evo_c[name], dims)


@pytest.mark.parametrize("path", g.case_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_var_types_match_reference(path):
    z, mats, dims, N_p, N_t = g.load_case(path)
    mask = cn.var_types(dims, nu_l=dims["nu_l"], nmu_l=dims["nmu_l"])
    ref = np.array([t == "b" for t in z["var_type_v"]])
    assert mask.shape == ref.shape and np.array_equal(mask, ref)
    assert int(mask.sum()) == dims["nv_l"]


def test_block_toeplitz_and_block_diag_match_reference():
    z = np.load(os.path.join(g.GDIR, "ref_matrix_utils.npz"))
    blocks = list(z["blocks"])
    assert np.array_equal(cn.block_toeplitz_lower(blocks), z["toeplitz"])
    assert np.array_equal(cn.block_diag_rep(blocks[0], 3), z["diag_small"])


def test_objective_weights_match_reference():
    z = np.load(os.path.join(g.GDIR, "ref_objective_weights.npz"))
    dims = dict(nx=1, nu=1, ndelta=0, nz=0, nmu=2, nomega=1, ny=1, nc=2, nv=3)
    tags = sorted({k.split("|")[0] for k in z.files if "|" in k and not k.startswith("spec")})
    seen = 0
    for tag in tags:
        spec = {str(k): z["specval_%s|%s" % (tag, k)] for k in z["spec_" + tag]}
        w = cn.build_weights(spec, dims, 4, 5)
        ref_keys = [k for k in z.files if k.startswith(tag + "|")]
        assert len(w) == len(ref_keys), (tag, list(w), ref_keys)
        for (var, atype, wtype, rate), val in w.items():
            key = "%s|%s|%s_%s%s" % (tag, var, atype, wtype, "_d" if rate else "")
            assert np.allclose(val, z[key], rtol=0, atol=1e-14), key
            seen += 1
    assert seen >= 10


# ---- time-varying horizons: condense_tv ---------------------------------------------------------------
@pytest.mark.parametrize("path", g.case_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_condense_tv_with_identical_steps_reproduces_reference_golden(path):
    z, mats, dims, N_p, N_t = g.load_case(path)
    evo = cn.condense_tv([mats] * N_t)
    for name in g.EVO_NAMES:
        g.check_evo(z, name, evo[name], dims)


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3"])
def test_condense_tv_is_the_step_by_step_evolution(name):
    """the condensed maps of a time-varying horizon against the recursion they summarise, simulated step by step"""
    import _tv
    from pyhybridcontrol_amd import synthetic as syn
    wl = syn.make_workload(name, batch=1)
    ag = wl["agents"][0]
    N = wl["N_tilde"]
    ms = _tv.step_models(ag["mats"], N, seed=3, strength=0.2)
    evo = cn.condense_tv(ms)
    d = evo["dims"]
    rng = np.random.default_rng(5)
    x0 = rng.normal(size=d["nx"]); V = rng.normal(size=(N, d["nv"])); W = rng.normal(size=(N, d["nomega"]))
    xs, ys, res = _tv.simulate(ms, d, x0, V, W)
    v, w, x = V.reshape(-1, 1), W.reshape(-1, 1), x0.reshape(-1, 1)
    sx = evo["Phi_x"] @ x + evo["Gamma_v"] @ v + evo["Gamma_omega"] @ w + evo["Gamma_5"]
    sy = evo["L_x"] @ x + evo["L_v"] @ v + evo["L_omega"] @ w + evo["L_5"]
    sr = evo["H_v"] @ v - (evo["H_x"] @ x + evo["H_omega"] @ w + evo["H_5"])
    for a, b in ((sx, xs), (sy, ys), (sr, res)):
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
    assert np.abs(evo["Gamma_v"] - cn.condense(ag["mats"], N)["Gamma_v"]).max() > 1e-3      # the horizon really varies
