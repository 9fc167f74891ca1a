import sys, numpy as np
sys.path.insert(0,'.')
from pyhybridcontrol_amd import _lib
from pyhybridcontrol_amd.batch import RcclGather, gather_sharded
_lib.check(_lib.load().mld_set_device(0))
uid = RcclGather.unique_id()
g = RcclGather(1, 0, uid)
a = np.arange(12, dtype=np.float64).reshape(6, 2)
out = gather_sharded(a, 6, 0, 1, g)
assert np.array_equal(out, a), out
print("RCCL all-gather (world=1) OK")
_lib.load().mld_comm_destroy()
