"""GPU parity for kernels K1+K2 (condensing) through the C ABI: HIP result vs (a) golden vectors made by
the reference itself and (b) the numpy oracle, on the same inputs.  fp64, tolerance 1e-11 relative."""
import os

import numpy as np
import pytest

import _golden as g
import condense_np as cn
from pyhybridcontrol_amd import gpu, synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("per_block", [False, True], ids=["k1-model", "k1-block"])
@pytest.mark.parametrize("path", g.case_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_condense_gpu_matches_reference_golden(path, per_block, monkeypatch):
    """both K1 variants: k_condense_model (all blocks of a model in one workgroup) and k_condense_blocks (kept for shapes
    that do not fit LDS)"""
    if per_block:
        monkeypatch.setenv("MLD_K1_PER_BLOCK", "1")
    z, mats, dims, N_p, N_t = g.load_case(path)
    m = gpu.GpuModel([mats], dims)
    evo = m.condense(N_t)
    for name in g.EVO_NAMES:
        g.check_evo(z, name, evo[name][0], dims)
    m.close()


def test_condense_gpu_batched_models_match_oracle():
    wl = syn.make_workload("cfg3", batch=1, n_agents=5)
    dims = wl["agents"][0]["dims"]
    m = gpu.GpuModel([a["mats"] for a in wl["agents"]], dims)
    evo = m.condense(wl["N_tilde"])
    for i, a in enumerate(wl["agents"]):
        ref = cn.condense(a["mats"], wl["N_tilde"])
        for name in g.EVO_NAMES:
            scale = max(1.0, float(np.abs(ref[name]).max()))
            assert np.abs(evo[name][i] - ref[name]).max() <= 1e-11 * scale, (i, name)
    # block-Toeplitz property at full size: shifting by one block row/column leaves H_v unchanged
    nc, nv = dims["nc"], m.nv
    Hv = evo["H_v"][0]
    assert np.array_equal(Hv[nc:, nv:], Hv[:-nc, :-nv])
    assert not np.any(np.triu(np.ones((wl["N_tilde"], wl["N_tilde"])), 1).repeat(nc, 0).repeat(nv, 1) * Hv)
    ms = m.condense_device(wl["N_tilde"])
    assert ms > 0
    m.close()


@pytest.mark.parametrize("path", g.case_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_fp32_materialisation_is_the_rounded_fp64_result(path):
    """mld_condense_f32 (MLD_F32 condensing, BASELINE configs[4]): every element is the fp64 value rounded once to fp32 -- bit-equal to
    the fp64 output cast to float32 -- and within 1e-5 (relative to the matrix scale) of the reference-made golden fixtures"""
    z, mats, dims, N_p, N_tilde = g.load_case(path)
    m = gpu.GpuModel([mats], dims)
    e64 = m.condense(N_tilde)
    e32 = m.condense(N_tilde, dtype=np.float32)
    for name in g.EVO_NAMES:
        assert e32[name].dtype == np.float32 and e32[name].shape == e64[name].shape
        assert np.array_equal(e32[name], e64[name].astype(np.float32)), name
        g.check_evo(z, name, e32[name][0].astype(np.float64), dims, rtol=1e-5)
    m.close()
