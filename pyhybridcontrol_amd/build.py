"""Build libmldgpu.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m pyhybridcontrol_amd.build
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mldgpu.hip")
OUT = os.environ.get("MLD_OUT") or os.path.join(HERE, "libmldgpu.so")      # (MLD_OUT + MLD_CXXFLAGS: a diagnostic variant beside the product, e.g. -DMLD_ASSERT)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    csrc = os.path.join(HERE, "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(HERE, "..", "include", "mldgpu.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
           "-Wno-unused-result"] + os.environ.get("MLD_CXXFLAGS", "").split() + ["-o", OUT, SRC, "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
