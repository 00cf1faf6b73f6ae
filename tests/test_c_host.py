"""A C99 host compiled with gcc against include/*.h and linked with the shared library -- the way the
reference is consumed.  CPU part: buffer.h ring KATs (test/test-structs.c:168-214) and the
no-device behaviour; -m gpu part: test/test-harmonic.c:32-48 and an analyze -> synthesize -> llsmrt loop in C."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "libllsm2_amd")
CFLAGS = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1", "-I" + INC]


def build_host(tmp):
    import libllsm2_amd
    libllsm2_amd.load()                                  # builds the .so if missing
    exe = os.path.join(tmp, "host_main")
    subprocess.check_call(CFLAGS + ["-o", exe, os.path.join(HERE, "c_host", "host_main.c"),
                                    "-L" + LIBDIR, "-l:libllsm2_amd.so", "-Wl,-rpath," + LIBDIR, "-lm"])
    return exe


def test_buffer_h_ring_kats(tmp_path):
    exe = str(tmp_path / "test_buffer")
    subprocess.check_call(CFLAGS + ["-o", exe, os.path.join(HERE, "c_host", "test_buffer.c")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "KATs ok" in out.stdout, out.stderr


def test_every_installed_header_compiles_as_c99(tmp_path):
    src = tmp_path / "all.c"
    src.write_text('#include "llsm.h"\n#include "llsmrt.h"\n#include "dsputils.h"\n#include "llsmutils.h"\n'
                   '#include "buffer.h"\n#include "llsm_gpu.h"\nint main(void) { return 0; }\n')
    subprocess.check_call(CFLAGS + ["-c", "-o", str(tmp_path / "all.o"), str(src)])


def test_a_double_build_of_a_host_is_refused_at_compile_time(tmp_path):
    """The reference can be built with FP_TYPE=double (makefile:20); this library is float only, and a host compiled with
    -DFP_TYPE=double against include/llsm.h must fail to compile instead of exchanging arrays of the wrong element size."""
    src = tmp_path / "dbl.c"
    src.write_text('#include "llsm.h"\nint main(void) { return 0; }\n')
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        out = subprocess.run([cc, std, "-DFP_TYPE=double", "-I" + INC, "-c", "-x", "c" if cc == "gcc" else "c++",
                              "-o", str(tmp_path / "dbl.o"), str(src)], capture_output=True, text=True)
        assert out.returncode != 0 and "llsm_amd_FP_TYPE_must_be_float" in out.stderr, (cc, out.stderr)
        ok = subprocess.run([cc, std, "-DFP_TYPE=float", "-I" + INC, "-c", "-x", "c" if cc == "gcc" else "c++",
                             "-o", str(tmp_path / "flt.o"), str(src)], capture_output=True, text=True)
        assert ok.returncode == 0, (cc, ok.stderr)


def test_c_host_without_device(tmp_path):
    exe = build_host(str(tmp_path))
    out = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "data model ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("slabs", [None, "1", "0"])
def test_c_host_on_gpu(tmp_path, slabs):
    """slabs: $LLSM_FRAME_SLABS unset (llsm_analyze returns heap frames, llsm_analyze_batch slab frames), 1 (slabs from
    llsm_analyze too: the round-3 default) and 0 (never)"""
    exe = build_host(str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k != "LLSM_FRAME_SLABS"}
    if slabs is not None:
        env["LLSM_FRAME_SLABS"] = slabs
    out = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300, env=env)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ICZT vs sinusoid bank" in out.stdout and "llsmrt" in out.stdout


@pytest.mark.gpu
def test_c_time_stretch_host_on_gpu(tmp_path):
    """tests/c_host/stretch_host.c: a C99 host that edits a model through the container API alone (the workflow of
    the reference's demo program) -- twice the frames by copying and blending, layer 1 -> layer 0, phase propagation,
    synthesis -- and checks length, level, long-term spectrum and F0 of the result."""
    import libllsm2_amd
    libllsm2_amd.load()
    exe = str(tmp_path / "stretch_host")
    subprocess.check_call(CFLAGS + ["-o", exe, os.path.join(HERE, "c_host", "stretch_host.c"),
                                    "-L" + LIBDIR, "-l:libllsm2_amd.so", "-Wl,-rpath," + LIBDIR, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "stretch ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_host_may_realloc_member_arrays_of_analysed_frames(tmp_path):
    """tests/c_host/realloc_host.c: the drop-in llsm_analyze returns ordinary heap frames by default (the reference's
    ownership rule; frame slabs are the additive llsm_analyze_batch's): a C host reallocs hm->ampl / eenv->ampl and frees
    nm->psd of every analysed frame, synthesises, deletes the chunk -- under glibc's heap checking, which aborts on a
    realloc / free of anything that is not the start of a heap block."""
    import libllsm2_amd
    libllsm2_amd.load()
    exe = str(tmp_path / "realloc_host")
    subprocess.check_call(CFLAGS + ["-o", exe, os.path.join(HERE, "c_host", "realloc_host.c"),
                                    "-L" + LIBDIR, "-l:libllsm2_amd.so", "-Wl,-rpath," + LIBDIR, "-lm"])
    env = {k: v for k, v in os.environ.items() if k != "LLSM_FRAME_SLABS"}
    env.update(MALLOC_CHECK_="3", MALLOC_PERTURB_="165")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    print(out.stdout)
    assert out.returncode == 0 and "realloc_host ok" in out.stdout, out.stdout + out.stderr


def test_frame_slabs_under_asan(tmp_path):
    """tests/c_host/slab_host.c with libllsm2_amd/csrc/model.cpp, both under -fsanitize=address,undefined (host code
    only: no device library in the process): frames carved out of one slab per chunk are copied, edited in place beyond
    their slab arrays, have members replaced / removed / attached past the container's size, are deleted singly and as a
    chunk; no invalid free, no leak, no slab left."""
    csrc = os.path.join(LIBDIR, "csrc")
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-O1", "-g"]
    mo, ho, exe = str(tmp_path / "model.o"), str(tmp_path / "slab_host.o"), str(tmp_path / "slab_host")
    subprocess.check_call(["g++", "-std=c++17"] + san + ["-I" + INC, "-I" + csrc, "-c", os.path.join(csrc, "model.cpp"), "-o", mo])
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror"] + san +
                          ["-I" + INC, "-c", os.path.join(HERE, "c_host", "slab_host.c"), "-o", ho])
    subprocess.check_call(["g++"] + san + [mo, ho, "-o", exe, "-lpthread"])
    for env in ({}, {"LLSM_SLAB_POOL_MB": "0"}):
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
        assert out.returncode == 0 and "slab_host ok" in out.stdout, (env, out.stdout + out.stderr)


def test_frames_over_packed_records_under_asan(tmp_path):
    """tests/c_host/packed_host.cpp with model.cpp under -fsanitize=address,undefined (host code only; the records a device
    would write are written by hand in csrc/packed.h's layout): the reference's accessors read the records' values, the
    synthesis-side view (llsm_chunk_packed_view) is granted while the frames lie untouched and withdrawn after any change of
    structure, copies outlive the chunk, every way of deleting leaves no slab behind -- three layouts, with and without the pool."""
    csrc = os.path.join(LIBDIR, "csrc")
    exe = str(tmp_path / "packed_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I" + INC, "-I" + csrc, os.path.join(csrc, "model.cpp"), os.path.join(HERE, "c_host", "packed_host.cpp"),
                           "-o", exe, "-lpthread"])
    for env in ({}, {"LLSM_SLAB_POOL_MB": "0"}, {"LLSM_SLAB_POOL_MB": "64"}):
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
        assert out.returncode == 0 and "packed_host ok" in out.stdout, (env, out.stdout + out.stderr)


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_frame_slabs_across_threads(tmp_path, sanitizer):
    """tests/c_host/slab_threads.cpp with model.cpp under -fsanitize=thread / address: eight threads build chunks, copy
    frames, grow members in place, replace and delete frames, and delete chunks that another thread built -- the slab
    registry, its per-thread cache and the pool are shared; no race, no invalid free, no slab left."""
    csrc = os.path.join(LIBDIR, "csrc")
    exe = str(tmp_path / ("slab_threads_" + sanitizer))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-I" + INC, "-I" + csrc,
                           os.path.join(csrc, "model.cpp"), os.path.join(HERE, "c_host", "slab_threads.cpp"), "-o", exe, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "slab_threads ok" in out.stdout and "WARNING: ThreadSanitizer" not in out.stderr, out.stdout + out.stderr


def test_lf_alpha_solve_against_bisection(tmp_path):
    """tests/c_host/lf_solve_check.cpp: lfmodel.h's alpha solve (sign scan on a table of e^-k, then safeguarded Newton steps;
    shared by the llsmrt pulse tracker on the host and the pulse kernels on the device) finds the bracket and the root of
    a plain scan-and-bisect reference: 1e-13 relative on the Rd curve, 1e-11 over random LF shapes."""
    exe = str(tmp_path / "lf_solve_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(LIBDIR, "csrc"),
                           os.path.join(HERE, "c_host", "lf_solve_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "0 bad" in out.stdout, out.stdout + out.stderr
