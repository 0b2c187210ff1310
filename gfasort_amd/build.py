"""In-tree build of libgfasort_hip.so (hipcc, gfx950 only).  The .so is git-ignored but travels
to the GPU box with the snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgfasort_hip.so")
SOURCES = ["sgd_kernels_1d.hip", "sgd_kernels_nd.hip", "sgd_kernels_nd_team.hip", "index_kernels.hip", "capi.hip", "multi.hip"]
HEADERS = ["sgd_device.h", "sgd_kernel_common.h", os.path.join("..", "..", "include", "gfasort_hip.h")]

# -ffp-contract=off : the reference (Rust) never fuses a*b+c; device and host tables must match it
# -munsafe-fp-atomics: native global_atomic_add_f64 on hipMalloc'ed (coarse-grained) memory
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
OBJDIR = os.path.join(HERE, "lib", "obj")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        sp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), hdr_t):
            return obj
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", "-o", obj, sp]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as ex:     # one TU per kernel family
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


HOST_DIR = os.path.join(CSRC, "host")
HOST_SOURCES = ["graph.cpp", "sgd.cpp"]
HOST_HEADERS = ["graph.hpp", "sgd.hpp"]
BINDIR = os.path.join(HERE, "bin")
CLI = os.path.join(BINDIR, "gfasort_hip")
SELFTEST = os.path.join(BINDIR, "host_selftest")
MULTI_SELFTEST = os.path.join(BINDIR, "multi_rank_selftest")
CXX_FLAGS = ["-O2", "-std=c++17", "-Wall", "-ffp-contract=off"]


def build_host(force=False, verbose=False):
    """The C++ host mirror (graph / GFA / params / Layout / SGD wrappers), its CLI `gfasort_hip`
    and `host_selftest`; both link libgfasort_hip.so through an $ORIGIN-relative rpath."""
    build_hip(force=False, verbose=verbose)
    os.makedirs(BINDIR, exist_ok=True)
    deps = [os.path.join(HOST_DIR, f) for f in HOST_SOURCES + HOST_HEADERS + ["main.cpp", "selftest.cpp"]] + [LIB]
    out = []
    for exe, main in ((CLI, "main.cpp"), (SELFTEST, "selftest.cpp")):
        if force or not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
            cmd = [os.environ.get("CXX", "g++")] + CXX_FLAGS + ["-o", exe, os.path.join(HOST_DIR, main)] + \
                  [os.path.join(HOST_DIR, f) for f in HOST_SOURCES] + \
                  ["-L" + LIBDIR, "-lgfasort_hip", "-pthread", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        out.append(exe)
    # multi_rank_selftest drives gfs_rank_* from one host thread per rank and stages its all-reduce through hipMemcpy: hipcc
    src = os.path.join(HOST_DIR, "multi_selftest.cpp")
    if force or not os.path.exists(MULTI_SELFTEST) or any(os.path.getmtime(d) > os.path.getmtime(MULTI_SELFTEST) for d in (src, LIB)):
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", MULTI_SELFTEST, src,
               "-L" + LIBDIR, "-lgfasort_hip", "-pthread", "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    out.append(MULTI_SELFTEST)
    return out


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
