"""pyhybridcontrol_amd: MI355X-native batched MLD-MPC solve path behind pyhybridcontrol's controller API.

Product path = libmldgpu.so (hand-written HIP for gfx950, C ABI in include/mldgpu.h) + this thin
Python host layer.  There is no CPU fallback: compute entry points raise MldGpuError without a GPU.
"""
from ._lib import MldGpuError, device_count, version  # noqa: F401
from .mld_model import (MldModel, MldInfo, ParNotSet, gen_schedule_params_tilde,  # noqa: F401
                        get_mld_numeric_tilde)
from .objective_atoms import ObjectiveAtoms  # noqa: F401
from .controllers import (MpcController, MldEvoMatrices, ControllerBuildRequiredError,  # noqa: F401
                          ControllerSolverError)
from .gpu import GpuModel, GpuProblem  # noqa: F401
from .batch import BatchSolver, shard_range, gather_sharded  # noqa: F401
from .aux_resolve import AuxResolver  # noqa: F401
from .compose import fuse  # noqa: F401
