"""Builds libllsm2_amd.so (HIP kernels for gfx950 + C-ABI host code) in-tree.

    python -m libllsm2_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but
travels with the tree to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libllsm2_amd.so")
SOURCES = ["kernels.hip", "l1_kernels.hip", "frame_kernels.hip", "frameapi.cpp", "coder.cpp", "l1.cpp", "engine.cpp", "capi.cpp", "model.cpp", "rt.cpp", "wire.cpp"]
HEADERS = ["kernels.h", "dev_common.h", "lfmodel.h", "batch.h", "scratch.h", "wave_fft.h", "engine.h", "plan.h", "cheby.h",
           os.path.join(ROOT, "include", "llsm.h"), os.path.join(ROOT, "include", "llsmrt.h"),
           os.path.join(ROOT, "include", "llsm_gpu.h"), os.path.join(ROOT, "include", "dsputils.h"),
           os.path.join(ROOT, "include", "llsmutils.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False, defines=(), out=None):
    """defines / out: experiment builds (tools/kbench.py ablations); the product build uses neither."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
           "-pthread", "-Wno-unused-result", "-Wno-unused-value",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out or LIB]
    cmd += ["-D" + d for d in defines]
    cmd += [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
