"""CPU only: after scripts/gpu_cl_repro.py died, read the trace and keep the inputs of the instances that were in flight -> gpurun_out/cl_candidates.npz"""
import numpy as np
tr = np.fromfile("gpurun_out/cl_trace.bin", np.int32).reshape(-1, 16)
import os
z = np.load("/tmp/cl_sub.npz") if os.path.exists("/tmp/cl_sub.npz") else np.load("/tmp/cl_in.npz")
print("source:", "hand-off sub-batch" if "fix" in z.files else "closed-loop step")
live = (tr[:, 0] >= 0) & (tr[:, 1] != 999)
inst = tr[live, 0]
print("workgroups in flight", int(live.sum()))
for row in tr[live][:300]: print("  inst %6d stage %5d pivots %6d queue pos %6d" % (row[0], row[1], row[2], row[3]))
extra = dict(fix=z["fix"][inst], cutoff=z["cutoff"][inst], round=z["round"]) if "fix" in z.files else dict(step=z["step"])
np.savez("gpurun_out/cl_candidates.npz", inst=inst, stage=tr[live, 1], pivots=tr[live, 2], x0=z["x0"][inst], omega=z["omega"][inst], midx=z["midx"][inst], **extra)
