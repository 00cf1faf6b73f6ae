"""Builds libllsm2_amd.so (HIP kernels for gfx950 + C-ABI host code) in-tree.

    python -m libllsm2_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but
travels with the tree to the GPU box.  Every source is compiled to its own
object (in parallel, cached under libllsm2_amd/_obj/) and the objects are
linked into one shared library.  The cache is keyed by CONTENT -- sha1 of the
source, of every header, of the flags and defines -- not by modification
times: a tree restored from a snapshot (round 6: rt.cpp came back with an
older mtime than an object compiled from a later draft of it) linked a stale
object into the product and hung every llsmrt feed on the GPU box.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libllsm2_amd.so")
OBJ = os.path.join(HERE, "_obj")
SOURCES = ["kernels.hip", "synth_kernels.hip", "l1_kernels.hip", "frame_kernels.hip", "frameapi.cpp", "coder.cpp", "l1.cpp",
           "engine.cpp", "capi.cpp", "model.cpp", "rt.cpp", "wire.cpp"]
import glob
# every header a source may include: csrc/*.h (launch.h was missing from a hand-kept list: ADVICE r4), include/*.h, and the
# timing experiments' header, which only -DLLSM_KBENCH_EXPERIMENTS builds include
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))) + \
          [os.path.join(ROOT, "tools", "kbench_experiments.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-slp-vectorize", "-pthread", "-Wno-unused-result", "-Wno-unused-value"]


def _headers():
    return [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]


def _sha(paths, extra=""):
    h = hashlib.sha1(extra.encode())
    for p in paths:
        if os.path.exists(p):
            h.update(p.encode() + b"\0")
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def _header_key():
    return _sha(_headers() + [os.path.abspath(__file__)], " ".join(FLAGS))


def _lib_key(defines=()):
    return _sha([os.path.join(CSRC, s) for s in SOURCES], _header_key() + " " + " ".join(sorted(defines)))


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _stale():
    return not os.path.exists(LIB) or _read(LIB + ".key") != _lib_key()


def build(force=False, verbose=False, defines=(), out=None):
    """defines / out: experiment builds (tools/kbench.py ablations); the product build uses neither."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = hashlib.sha1(" ".join(sorted(defines)).encode()).hexdigest()[:10] if defines else "product"
    odir = os.path.join(OBJ, tag)
    os.makedirs(odir, exist_ok=True)
    base = [hipcc] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + ["-D" + d for d in defines]
    hdr_key = _header_key()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def one(s):
        src = os.path.join(CSRC, s)
        obj = os.path.join(odir, s + ".o")
        key = _sha([src], hdr_key + " " + " ".join(sorted(defines)))
        if not force and os.path.exists(obj) and _read(obj + ".key") == key:
            return obj
        cmd = base + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(obj + ".key", "w") as f:
            f.write(key)
        return obj

    with ThreadPoolExecutor(max(1, min(len(srcs), os.cpu_count() or 4))) as ex:
        objs = list(ex.map(one, srcs))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out or LIB] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    with open((out or LIB) + ".key", "w") as f:
        f.write(_lib_key(defines))
    return out or LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
