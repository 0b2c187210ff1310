"""ctypes loader for the CPU oracle (oracle/gfs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under gfasort_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgfs_oracle.so")

NO_NODE = 0xFFFFFFFF


class GfoGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_steps", C.c_uint64), ("n_paths", C.c_uint64),
                ("node_len", C.c_void_p), ("step_node", C.c_void_p),
                ("step_is_rev", C.c_void_p), ("path_first_step", C.c_void_p)]


class GfoParams(C.Structure):
    _fields_ = [("iter_max", C.c_uint64), ("iter_with_max_learning_rate", C.c_uint64),
                ("min_term_updates", C.c_uint64), ("delta", C.c_double), ("eps", C.c_double),
                ("eta_max", C.c_double), ("theta", C.c_double), ("space", C.c_uint64),
                ("space_max", C.c_uint64), ("space_quantization_step", C.c_uint64),
                ("cooling_start", C.c_double), ("nthreads", C.c_uint64), ("seed", C.c_uint64),
                ("dimensions", C.c_uint64)]


class GfoStats(C.Structure):
    _fields_ = [("term_updates", C.c_uint64), ("attempts", C.c_uint64),
                ("iterations", C.c_uint64), ("seconds", C.c_double)]


TERM_DTYPE = np.dtype([("i", "<u4"), ("j", "<u4"), ("d_ij", "<f8")], align=True)


def build(force=False):
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "gfs_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.gfo_fast_precise_pow.restype = C.c_double
        L.gfo_fast_precise_pow.argtypes = [C.c_double, C.c_double]
        L.gfo_dirty_zipfian.restype = C.c_uint64
        L.gfo_dirty_zipfian.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double]
        L.gfo_splitmix64_next.restype = C.c_uint64
        L.gfo_splitmix64_next.argtypes = [C.POINTER(C.c_uint64)]
        L.gfo_xoshiro_seed.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
        L.gfo_xoshiro_next.restype = C.c_uint64
        L.gfo_xoshiro_next.argtypes = [C.POINTER(C.c_uint64)]
        L.gfo_uniform_usize.restype = C.c_uint64
        L.gfo_uniform_usize.argtypes = [C.POINTER(C.c_uint64), C.c_uint64]
        L.gfo_flip.restype = C.c_uint32
        L.gfo_flip.argtypes = [C.POINTER(C.c_uint64)]
        L.gfo_random_f64.restype = C.c_double
        L.gfo_random_f64.argtypes = [C.POINTER(C.c_uint64)]
        L.gfo_zeta_size.restype = C.c_uint64
        L.gfo_layout_stress.restype = C.c_double
        L.gfo_layout_stress.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.gfo_stress_1d.restype = C.c_double
        L.gfo_stress_1d.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Graph:
    """Keeps the numpy arrays alive behind a gfo_graph view."""

    def __init__(self, node_len, step_node, step_is_rev, path_first_step):
        self.node_len = np.ascontiguousarray(node_len, dtype=np.uint32)
        self.step_node = np.ascontiguousarray(step_node, dtype=np.uint32)
        self.step_is_rev = np.ascontiguousarray(step_is_rev, dtype=np.uint8)
        self.path_first_step = np.ascontiguousarray(path_first_step, dtype=np.uint64)
        self.c = GfoGraph(len(self.node_len), len(self.step_node), len(self.path_first_step) - 1,
                          _ptr(self.node_len), _ptr(self.step_node), _ptr(self.step_is_rev),
                          _ptr(self.path_first_step))

    @property
    def ref(self):
        return C.byref(self.c)


def params(**kw):
    p = GfoParams()
    defaults = dict(iter_max=100, iter_with_max_learning_rate=0, min_term_updates=100, delta=0.0,
                    eps=0.01, eta_max=100.0, theta=0.99, space=100, space_max=100,
                    space_quantization_step=100, cooling_start=0.5, nthreads=1, seed=9399220,
                    dimensions=2)
    defaults.update(kw)
    for k, v in defaults.items():
        setattr(p, k, v)
    return p


def fast_precise_pow(a, b):
    return lib().gfo_fast_precise_pow(a, b)


def dirty_zipfian(mn, mx, theta, zeta, zeta2theta, u):
    return lib().gfo_dirty_zipfian(mn, mx, theta, zeta, zeta2theta, u)


class Xoshiro:
    def __init__(self, seed=None, state=None):
        self.s = (C.c_uint64 * 4)()
        if state is not None:
            for k in range(4):
                self.s[k] = state[k]
        else:
            lib().gfo_xoshiro_seed(seed, self.s)

    def next_u64(self):
        return lib().gfo_xoshiro_next(self.s)

    def uniform(self, n):
        return lib().gfo_uniform_usize(self.s, n)

    def flip(self):
        return lib().gfo_flip(self.s)

    def random_f64(self):
        return lib().gfo_random_f64(self.s)

    def state(self):
        return [int(v) for v in self.s]


def splitmix64_stream(seed, n):
    st = C.c_uint64(seed)
    return [lib().gfo_splitmix64_next(C.byref(st)) for _ in range(n)]


def schedule(p):
    etas = np.zeros(p.iter_max + 1, dtype=np.float64)
    lib().gfo_schedule(C.byref(p), _ptr(etas))
    return etas


def zetas(p):
    n = lib().gfo_zeta_size(C.byref(p))
    z = np.zeros(n, dtype=np.float64)
    lib().gfo_zetas(C.byref(p), _ptr(z))
    return z


def path_index(g):
    S, P = g.c.n_steps, g.c.n_paths
    pos = np.zeros(S, dtype=np.uint64)
    pth = np.zeros(S, dtype=np.uint32)
    rank = np.zeros(S, dtype=np.uint64)
    plen = np.zeros(P, dtype=np.uint64)
    lib().gfo_path_index(g.ref, _ptr(pos), _ptr(pth), _ptr(rank), _ptr(plen))
    return pos, pth, rank, plen


def init_positions(g):
    x = np.zeros(g.c.n_nodes, dtype=np.float64)
    lib().gfo_init_positions(g.ref, _ptr(x))
    return x


def init_layout(g, dims, seed):
    """sgd.rs:829-853 with the restated rand_distr StandardNormal (parity unpinned)."""
    c = np.zeros(len(g.node_len) * 2 * dims, dtype=np.float64)
    lib().gfo_init_layout(g.ref, C.c_uint64(dims), C.c_uint64(seed), _ptr(c))
    return c


def ziggurat_tables():
    x, f = np.zeros(257), np.zeros(257)
    lib().gfo_ziggurat_tables(_ptr(x), _ptr(f))
    return x, f


def standard_normal(seed, n):
    rng = (C.c_uint64 * 4)()
    lib().gfo_xoshiro_seed(seed, rng)
    lib().gfo_standard_normal.restype = C.c_double
    return np.array([lib().gfo_standard_normal(rng) for _ in range(n)])


def init_layout_dim0(g, dims, coords=None):
    if coords is None:
        coords = np.zeros(g.c.n_nodes * 2 * dims, dtype=np.float64)
    lib().gfo_init_layout_dim0(g.ref, C.c_uint64(dims), _ptr(coords))
    return coords


def _run(fn, g, p, x, n_streams, attempt_factor, trace_per_stream, etas, zts):
    st = GfoStats()
    trace = None
    if trace_per_stream:
        trace = np.zeros(n_streams * trace_per_stream, dtype=TERM_DTYPE)
    rc = fn(g.ref, C.byref(p), _ptr(etas), _ptr(zts), C.c_uint64(n_streams),
            C.c_uint64(attempt_factor), _ptr(x), _ptr(trace), C.c_uint64(trace_per_stream), C.byref(st))
    return rc, st, trace


def sgd_1d(g, p, x, n_streams=1, attempt_factor=64, trace_per_stream=0, etas=None, zts=None):
    """Deterministic-mode 1D SGD; x (float64[n_nodes]) is updated in place."""
    return _run(lib().gfo_sgd_1d, g, p, x, n_streams, attempt_factor, trace_per_stream, etas, zts)


def sgd_nd(g, p, coords, n_streams=1, attempt_factor=64, trace_per_stream=0, etas=None, zts=None):
    """Deterministic-mode nD SGD; coords (float64[n_nodes*2*D], Layout order) updated in place."""
    return _run(lib().gfo_sgd_nd, g, p, coords, n_streams, attempt_factor, trace_per_stream, etas, zts)


def sgd_1d_threads(g, p, x, flat=0, max_seconds=0.0, etas=None, zts=None):
    st = GfoStats()
    rc = lib().gfo_sgd_1d_threads(g.ref, C.byref(p), _ptr(etas), _ptr(zts), C.c_int(flat),
                                  C.c_double(max_seconds), _ptr(x), C.byref(st))
    return rc, st


def sgd_nd_threads(g, p, coords, flat=0, max_seconds=0.0, etas=None, zts=None):
    st = GfoStats()
    rc = lib().gfo_sgd_nd_threads(g.ref, C.byref(p), _ptr(etas), _ptr(zts), C.c_int(flat),
                                  C.c_double(max_seconds), _ptr(coords), C.byref(st))
    return rc, st


class State:
    """Resumable deterministic run (gfo_state).  bundle > 1 mirrors the product's bundled sampler."""

    def __init__(self, g, p, dims=0, n_streams=1, stream_base=0, quota_total=0, attempt_factor=64,
                 trace_per_stream=0, bundle=1, etas=None, zts=None, node_slots=None, one_colour=False, chain=1, fused_trip=True, partners=1, twin_trip=True, chunk=0):
        self.g, self.p = g, p
        self.trace = np.zeros(n_streams * trace_per_stream, dtype=TERM_DTYPE) if trace_per_stream else None
        self.h = C.c_void_p()
        rc = lib().gfo_state_create(g.ref, C.byref(p), _ptr(etas), _ptr(zts), C.c_uint64(dims), C.c_uint64(n_streams),
                                    C.c_uint64(stream_base), C.c_uint64(quota_total), C.c_uint64(attempt_factor),
                                    _ptr(self.trace), C.c_uint64(trace_per_stream), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"gfo_state_create rc={rc}")
        if bundle != 1 and lib().gfo_state_set_bundle(self.h, C.c_uint64(bundle)) != 0:
            raise ValueError("bad bundle")
        if one_colour:
            assert lib().gfo_state_set_one_colour(self.h, C.c_int(1)) == 0
        if partners != 1 or not twin_trip:
            # the product's two partners per leader (default of its 1D team kernel at B = 64; GFS_F_ONE_PARTNER = 1)
            assert lib().gfo_state_set_partners(self.h, C.c_int(partners), C.c_int(0 if twin_trip else 1)) == 0
        if not fused_trip:
            assert lib().gfo_state_set_no_fused_trip(self.h, C.c_int(1)) == 0       # mirror of GFS_F_DBG_NO_FUSED_TRIP
        if chunk:
            # updates per chunk of a team wave's work (the product shortens it for a pooled launch of ONE small iteration)
            assert lib().gfo_state_set_chunk(self.h, C.c_uint64(chunk)) == 0
        if chain != 1:
            # the product's long runs (GFS_F_CHAIN; its default at B = 64 is 64)
            assert lib().gfo_state_set_chain(self.h, C.c_uint64(chain)) == 0
        if node_slots is not None:
            # the product's internal node layout (hip.Context.node_layout()): its bundled sampler aligns runs to it
            slots = np.ascontiguousarray(node_slots, dtype=np.uint32)
            assert lib().gfo_state_set_node_slots(self.h, _ptr(slots)) == 0
        self.n_streams, self.trace_per_stream = n_streams, trace_per_stream

    def run_iteration(self, k, x):
        assert lib().gfo_state_run_iteration(self.h, C.c_uint64(k), _ptr(x)) == 0

    def run(self, x):
        for k in range(self.p.iter_max + 1):
            self.run_iteration(k, x)

    def stats(self):
        st = GfoStats()
        lib().gfo_state_stats(self.h, C.byref(st))
        return st

    def close(self):
        if self.h:
            lib().gfo_state_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def layout_stress(g, dims, coords, samples=10000):
    return lib().gfo_layout_stress(g.ref, dims, _ptr(np.ascontiguousarray(coords, dtype=np.float64)), samples)


def stress_1d(g, x, samples=10000):
    return lib().gfo_stress_1d(g.ref, _ptr(np.ascontiguousarray(x, dtype=np.float64)), samples)
