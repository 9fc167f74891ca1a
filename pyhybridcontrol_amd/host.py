"""Small host-side helpers shared by the controller mirror, the batch harness and the tests."""
import numpy as np

from .objective_atoms import ObjectiveAtoms


def cost_from_atoms(atoms, dims, N_p, N_tilde):
    """string-keyed objective atoms (reference syntax) -> tiled cost dict for gpu.GpuProblem"""
    oa = ObjectiveAtoms(dims, N_p, N_tilde, atoms)
    c = oa.to_cost()
    c.pop("_omega_atoms", None)
    return c


def stack_costs(cost_list):
    """list of per-model cost dicts -> one dict of (n_models, ...) arrays (None where nobody has a term)"""
    out = {}
    for k in ("lin_v", "lin_x", "lin_y", "quad_v", "quad_x", "quad_y"):
        vals = [c.get(k) for c in cost_list]
        if all(v is None for v in vals):
            out[k] = None
            continue
        ref = next(v for v in vals if v is not None)
        out[k] = np.stack([np.zeros_like(ref) if v is None else np.asarray(v, np.float64) for v in vals])
    return out
