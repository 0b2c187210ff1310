"""ctypes binding of libgfasort_hip.so (include/gfasort_hip.h).  There is no CPU fallback:
if the library is missing this module raises, and compute calls on a machine without a HIP
device return GFS_E_HIP (raised as GfsError)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

NO_NODE = 0xFFFFFFFF
MAX_DIMS = 8
OK, NOTHING_TO_DO = 0, 1
F_PLAIN_LOADS, F_NO_LDS_TABLES, F_NO_FUSE = 1, 2, 4


F_DBG_NO_ATOMICS, F_DBG_NO_XLOADS, F_DBG_ONE_COLOUR = 0x100, 0x200, 0x800
F_DBG_NO_ALIGN, F_DBG_ALIGN_FIRST, F_DBG_WIDE_INDEX, F_DBG_NO_FUSED_TRIP = 0x1000, 0x2000, 0x4000, 0x8000
F_ONE_PARTNER, F_DBG_NO_TWIN_TRIP, F_DBG_FREE_RUNNING = 0x10, 0x400, 0x20


def F_CHAIN(k):
    """GFS_F_CHAIN(k): longest run in trips at B = 64 (0 = auto: 64 for the 1D sort, 16 for layouts; else a power of two <= 64)."""
    return (int(k) & 0xFF) << 24


def F_BUNDLE(n):
    """GFS_F_BUNDLE(n): 0 = auto, 1 = reference streams, 4..64 = explicit bundle width."""
    return (int(n) & 0xFF) << 16


class GfsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gfasort_hip error {code}: {msg}")
        self.code = code


class GraphView(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_steps", C.c_uint64), ("n_paths", C.c_uint64),
                ("node_len", C.c_void_p), ("step_node", C.c_void_p),
                ("step_is_rev", C.c_void_p), ("path_first_step", C.c_void_p)]


class SgdParams(C.Structure):
    _fields_ = [("iter_max", C.c_uint64), ("iter_with_max_learning_rate", C.c_uint64),
                ("min_term_updates", C.c_uint64), ("delta", C.c_double), ("eps", C.c_double),
                ("eta_max", C.c_double), ("theta", C.c_double), ("space", C.c_uint64),
                ("space_max", C.c_uint64), ("space_quantization_step", C.c_uint64),
                ("cooling_start", C.c_double), ("nthreads", C.c_uint64),
                ("progress", C.c_uint8), ("_pad", C.c_uint8 * 7), ("seed", C.c_uint64)]


class LayoutParams(C.Structure):
    _fields_ = [("dimensions", C.c_uint64), ("sgd", SgdParams)]


class LaunchConfig(C.Structure):
    _fields_ = [("n_streams", C.c_uint64), ("stream_base", C.c_uint64),
                ("term_updates_per_iteration", C.c_uint64), ("attempt_factor", C.c_uint64),
                ("block_size", C.c_uint32), ("flags", C.c_uint32), ("trace_per_stream", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("term_updates", C.c_uint64), ("attempts", C.c_uint64), ("iterations", C.c_uint64),
                ("n_streams", C.c_uint64), ("bundle", C.c_uint64), ("kernel_ms", C.c_double),
                ("total_ms", C.c_double), ("launches", C.c_uint64), ("run_trips", C.c_uint64)]


class RankConfig(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("device", C.c_int32), ("sharding", C.c_uint32),
                ("merge_every", C.c_uint32), ("merge_rule", C.c_uint32), ("payload", C.c_uint32), ("exchange", C.c_uint32),
                ("launch", LaunchConfig)]


class RankInfo(C.Structure):
    _fields_ = [("quota", C.c_uint64), ("shard_steps", C.c_uint64), ("span_lo", C.c_uint64), ("span_hi", C.c_uint64),
                ("shared_slots", C.c_uint64), ("exchange_count", C.c_uint64), ("positions_len", C.c_uint64),
                ("windows", C.c_uint64), ("last_merge_kernels_ms", C.c_double), ("idle", C.c_uint32), ("_pad", C.c_uint32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p)
MERGE_RULES = {"anneal": 0, "sum": 1, "mean": 2, "touch": 3}

TERM_DTYPE = np.dtype([("i", "<u4"), ("j", "<u4"), ("d_ij", "<f8")], align=True)

# every symbol include/gfasort_hip.h declares
EXPORTS = [
    "gfs_version", "gfs_last_error", "gfs_device_count", "gfs_warmup", "gfs_fast_precise_pow", "gfs_sgd_schedule",
    "gfs_zeta_table_len", "gfs_zeta_table", "gfs_init_positions", "gfs_init_layout_dim0", "gfs_init_layout",
    "gfs_sort_order", "gfs_path_linear_sgd", "gfs_path_sgd_sort", "gfs_path_linear_sgd_layout", "gfs_ctx_create",
    "gfs_ctx_create_with_layout", "gfs_ctx_node_layout",
    "gfs_ctx_destroy", "gfs_ctx_setup_1d", "gfs_ctx_setup_nd", "gfs_ctx_positions_len", "gfs_ctx_init_positions",
    "gfs_ctx_upload_positions", "gfs_ctx_download_positions", "gfs_ctx_positions_device",
    "gfs_ctx_bind_positions", "gfs_ctx_reset_streams", "gfs_ctx_run_iteration", "gfs_ctx_run_range", "gfs_ctx_run",
    "gfs_ctx_synchronize", "gfs_ctx_stats", "gfs_ctx_sort_order", "gfs_ctx_trace", "gfs_merge_prepare", "gfs_merge_apply",
    "gfs_shard_paths", "gfs_shard_quotas", "gfs_shared_node_layout", "gfs_exchange_plan",
    "gfs_rank_create", "gfs_rank_destroy", "gfs_rank_ctx", "gfs_rank_get_info", "gfs_rank_set_positions",
    "gfs_rank_positions_changed", "gfs_rank_get_positions", "gfs_rank_exchange_count", "gfs_rank_exchange_buffer",
    "gfs_rank_bind_exchange_buffer", "gfs_rank_window_begin", "gfs_rank_window_end", "gfs_rank_finish_begin",
    "gfs_rank_finish_buffer", "gfs_rank_finish_end", "gfs_rank_run",
]

_lib = None


def lib_path():
    return _build.LIB


def lib():
    """Load libgfasort_hip.so; raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        path = os.environ.get("GFS_LIB_PATH") or _build.LIB      # (GFS_LIB_PATH: experiment builds, scripts/ only)
        if not os.path.exists(path):
            raise ImportError(f"{path} is not built: run `python -m gfasort_amd.build` "
                              "(hipcc, gfx950). There is no CPU fallback.")
        L = C.CDLL(path)
        L.gfs_version.restype = C.c_char_p
        L.gfs_last_error.restype = C.c_char_p
        L.gfs_fast_precise_pow.restype = C.c_double
        L.gfs_fast_precise_pow.argtypes = [C.c_double, C.c_double]
        L.gfs_zeta_table_len.restype = C.c_uint64
        L.gfs_ctx_positions_len.restype = C.c_uint64
        L.gfs_ctx_positions_len.argtypes = [C.c_void_p]
        L.gfs_ctx_init_positions.argtypes = [C.c_void_p]
        L.gfs_ctx_positions_device.restype = C.c_void_p
        L.gfs_ctx_positions_device.argtypes = [C.c_void_p]
        L.gfs_ctx_destroy.argtypes = [C.c_void_p]
        L.gfs_ctx_destroy.restype = None
        L.gfs_ctx_run_iteration.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.gfs_ctx_run.argtypes = [C.c_void_p, C.c_void_p]
        L.gfs_ctx_run_range.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.gfs_ctx_synchronize.argtypes = [C.c_void_p, C.c_void_p]
        L.gfs_ctx_bind_positions.argtypes = [C.c_void_p, C.c_void_p]
        L.gfs_ctx_reset_streams.argtypes = [C.c_void_p]
        L.gfs_ctx_upload_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.gfs_ctx_download_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.gfs_ctx_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.gfs_ctx_sort_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.gfs_ctx_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.gfs_ctx_node_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.gfs_ctx_setup_1d.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.gfs_ctx_setup_nd.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.gfs_sort_order.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.gfs_merge_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.gfs_merge_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p]
        L.gfs_shard_paths.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.gfs_shard_quotas.argtypes = [C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
        L.gfs_shared_node_layout.argtypes = [C.c_void_p, C.c_void_p]
        L.gfs_exchange_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 9
        L.gfs_rank_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.gfs_rank_destroy.argtypes = [C.c_void_p]
        L.gfs_rank_destroy.restype = None
        L.gfs_rank_ctx.argtypes = [C.c_void_p]
        L.gfs_rank_ctx.restype = C.c_void_p
        L.gfs_rank_get_info.argtypes = [C.c_void_p, C.c_void_p]
        L.gfs_rank_set_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.gfs_rank_positions_changed.argtypes = [C.c_void_p, C.c_void_p]
        L.gfs_rank_get_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.gfs_rank_exchange_count.argtypes = [C.c_void_p]
        L.gfs_rank_exchange_count.restype = C.c_uint64
        L.gfs_rank_exchange_buffer.argtypes = [C.c_void_p]
        L.gfs_rank_exchange_buffer.restype = C.c_void_p
        L.gfs_rank_bind_exchange_buffer.argtypes = [C.c_void_p, C.c_void_p]
        L.gfs_rank_window_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.gfs_rank_window_end.argtypes = [C.c_void_p, C.c_void_p]
        L.gfs_rank_finish_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.gfs_rank_finish_buffer.argtypes = [C.c_void_p]
        L.gfs_rank_finish_buffer.restype = C.c_void_p
        L.gfs_rank_finish_end.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.gfs_rank_run.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc < 0:
        raise GfsError(rc, lib().gfs_last_error().decode())
    return rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_view(g):
    """gfs_graph_view over a FlatGraph; returns (view, keepalive)."""
    arrs = (np.ascontiguousarray(g.node_len, dtype=np.uint32),
            np.ascontiguousarray(g.step_node, dtype=np.uint32),
            np.ascontiguousarray(g.step_is_rev, dtype=np.uint8),
            np.ascontiguousarray(g.path_first_step, dtype=np.uint64))
    v = GraphView(g.n_nodes, g.n_steps, g.n_paths, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(arrs[3]))
    return v, arrs


def make_sgd_params(p):
    s = SgdParams()
    for name in ("iter_max", "iter_with_max_learning_rate", "min_term_updates", "delta", "eps", "eta_max",
                 "theta", "space", "space_max", "space_quantization_step", "cooling_start", "nthreads", "seed"):
        setattr(s, name, getattr(p, name))
    s.progress = 1 if p.progress else 0
    return s


def make_layout_params(p):
    lp = LayoutParams()
    lp.dimensions = p.dimensions
    lp.sgd = make_sgd_params(p)
    return lp


def make_config(n_streams=0, stream_base=0, term_updates_per_iteration=0, attempt_factor=0,
                block_size=0, flags=0, trace_per_stream=0):
    return LaunchConfig(n_streams, stream_base, term_updates_per_iteration, attempt_factor,
                        block_size, flags, trace_per_stream)


# ---- host tables -----------------------------------------------------------------------------
def fast_precise_pow(a, b):
    return lib().gfs_fast_precise_pow(a, b)


def sgd_schedule(p):
    sp = make_sgd_params(p)
    etas = np.zeros(p.iter_max + 1, dtype=np.float64)
    check(lib().gfs_sgd_schedule(C.byref(sp), _ptr(etas)))
    return etas


def zeta_table(p):
    sp = make_sgd_params(p)
    n = lib().gfs_zeta_table_len(C.byref(sp))
    z = np.zeros(n, dtype=np.float64)
    check(lib().gfs_zeta_table(C.byref(sp), _ptr(z)))
    return z


def init_positions(g):
    v, keep = make_view(g)
    x = np.zeros(g.n_nodes, dtype=np.float64)
    check(lib().gfs_init_positions(C.byref(v), _ptr(x)))
    return x


def init_layout_dim0(g, dims, coords=None):
    v, keep = make_view(g)
    if coords is None:
        coords = np.zeros(g.n_nodes * 2 * dims, dtype=np.float64)
    check(lib().gfs_init_layout_dim0(C.byref(v), C.c_uint64(dims), _ptr(coords)))
    return coords


def init_layout(g, dims, seed):
    """gfs_init_layout: the reference's whole layout start (dimension 0 + StandardNormal dimensions), Layout order."""
    v, keep = make_view(g)
    coords = np.zeros(g.n_nodes * 2 * dims, dtype=np.float64)
    check(lib().gfs_init_layout(C.byref(v), C.c_uint64(dims), C.c_uint64(seed), _ptr(coords)))
    return coords


def sort_order(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    order = np.zeros(x.shape[0], dtype=np.uint64)
    check(lib().gfs_sort_order(_ptr(x), x.shape[0], _ptr(order)))
    return order


def merge_prepare(x_ptr, x_prev_ptr, buf_ptr, n, stream=None):
    check(lib().gfs_merge_prepare(C.c_void_p(x_ptr), C.c_void_p(x_prev_ptr), C.c_void_p(buf_ptr), C.c_uint64(n),
                                  C.c_void_p(stream or 0)))


def merge_apply(x_ptr, x_prev_ptr, buf_ptr, n, divide_all_by=0.0, stream=None):
    check(lib().gfs_merge_apply(C.c_void_p(x_ptr), C.c_void_p(x_prev_ptr), C.c_void_p(buf_ptr), C.c_uint64(n),
                                C.c_double(divide_all_by), C.c_void_p(stream or 0)))


# ---- multi-device planning (host only) and the rank object -------------------------------------
class ShardPlan:
    """gfs_shard_paths / gfs_shard_quotas / gfs_shared_node_layout / gfs_exchange_plan of a graph for `world` ranks."""

    def __init__(self, g, min_term_updates, world, sharding=0, whole_vector=False):
        v, keep = make_view(g)
        self.world = world
        self.path_owner = np.zeros(max(g.n_paths, 1), dtype=np.uint32)
        self.rank_steps = np.zeros(world, dtype=np.uint64)
        check(lib().gfs_shard_paths(C.byref(v), world, sharding, _ptr(self.path_owner), _ptr(self.rank_steps)))
        self.path_owner = self.path_owner[:g.n_paths]
        self.quotas = np.zeros(world, dtype=np.uint64)
        check(lib().gfs_shard_quotas(C.c_uint64(int(min_term_updates)), _ptr(self.rank_steps), world, _ptr(self.quotas)))
        self.perm = np.zeros(max(g.n_nodes, 1), dtype=np.uint32)
        check(lib().gfs_shared_node_layout(C.byref(v), _ptr(self.perm)))
        self.perm = self.perm[:g.n_nodes]
        self.span_lo, self.span_hi = np.zeros(world, dtype=np.uint64), np.zeros(world, dtype=np.uint64)
        seg_lo, seg_hi = np.zeros(2 * world, dtype=np.uint64), np.zeros(2 * world, dtype=np.uint64)
        own_lo, own_hi = np.zeros(2 * world + 2, dtype=np.uint64), np.zeros(2 * world + 2, dtype=np.uint64)
        own_rank = np.zeros(2 * world + 2, dtype=np.uint32)
        n_seg, n_own = C.c_uint32(0), C.c_uint32(0)
        po = np.ascontiguousarray(self.path_owner) if g.n_paths else np.zeros(1, dtype=np.uint32)
        pm = np.ascontiguousarray(self.perm) if g.n_nodes else np.zeros(1, dtype=np.uint32)
        check(lib().gfs_exchange_plan(C.byref(v), _ptr(pm), _ptr(po), world, _ptr(self.span_lo), _ptr(self.span_hi),
                                      _ptr(seg_lo), _ptr(seg_hi), C.byref(n_seg), _ptr(own_lo), _ptr(own_hi), _ptr(own_rank),
                                      C.byref(n_own)))
        self.shared = [(int(seg_lo[k]), int(seg_hi[k])) for k in range(n_seg.value)]      # slots two or more ranks can move
        if whole_vector and g.n_nodes:
            self.shared = [(0, g.n_nodes)]
        self.owned = [(int(own_lo[k]), int(own_hi[k]), int(own_rank[k])) for k in range(n_own.value)]

    def paths_of(self, rank):
        return np.flatnonzero(self.path_owner == rank).tolist()


class Rank:
    """gfs_rank: one rank of a multi-device run (its shard's context, the exchange of the shared slots)."""

    def __init__(self, g, p, dims, rank, world, device=0, sharding=0, merge_every=1, merge_rule="anneal", payload_f64=False,
                 whole_vector=False, launch=None):
        self.graph, self.params, self.dims = g, p, dims
        v, self._keep = make_view(g)
        cfg = RankConfig(rank, world, device, sharding, merge_every, MERGE_RULES[merge_rule], 1 if payload_f64 else 0,
                         1 if whole_vector else 0, launch if launch is not None else LaunchConfig())
        self.cfg = cfg
        sp = make_sgd_params(p)
        self._h = C.c_void_p()
        self.rc = check(lib().gfs_rank_create(C.byref(v), C.byref(sp), C.c_uint64(dims), C.byref(cfg), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().gfs_rank_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        out = RankInfo()
        check(lib().gfs_rank_get_info(self._h, C.byref(out)))
        return out

    def ctx_stats(self):
        st = Stats()
        check(lib().gfs_ctx_stats(C.c_void_p(lib().gfs_rank_ctx(self._h)), C.byref(st)))
        return st

    def reset_streams(self):
        check(lib().gfs_ctx_reset_streams(C.c_void_p(lib().gfs_rank_ctx(self._h))))

    def set_positions(self, x=None):
        if x is None:
            check(lib().gfs_rank_set_positions(self._h, None, 0))
        else:
            x = np.ascontiguousarray(x, dtype=np.float64)
            check(lib().gfs_rank_set_positions(self._h, _ptr(x), x.shape[0]))

    def get_positions(self):
        x = np.zeros(int(self.info().positions_len), dtype=np.float64)
        check(lib().gfs_rank_get_positions(self._h, _ptr(x), x.shape[0]))
        return x

    def exchange_count(self):
        return int(lib().gfs_rank_exchange_count(self._h))

    def bind_exchange_buffer(self, device_ptr):
        check(lib().gfs_rank_bind_exchange_buffer(self._h, C.c_void_p(device_ptr)))

    def window_begin(self, ks, stream=None):
        ks = np.ascontiguousarray(ks, dtype=np.uint64)
        return check(lib().gfs_rank_window_begin(self._h, _ptr(ks), C.c_uint64(ks.shape[0]), C.c_void_p(stream or 0)))

    def window_end(self, stream=None):
        check(lib().gfs_rank_window_end(self._h, C.c_void_p(stream or 0)))

    def finish_begin(self, full_device_ptr, stream=None):
        check(lib().gfs_rank_finish_begin(self._h, C.c_void_p(full_device_ptr), C.c_void_p(stream or 0)))

    def finish_end(self, full_device_ptr, stream=None):
        check(lib().gfs_rank_finish_end(self._h, C.c_void_p(full_device_ptr), C.c_void_p(stream or 0)))

    def run(self, allreduce=None, stream=None):
        """gfs_rank_run with a Python collective: allreduce(device_ptr, count, is_f64, stream) -> None."""
        def tramp(user, buf, count, is_f64, st):
            try:
                allreduce(buf, count, bool(is_f64), st)
                return 0
            except Exception:                                  # never raise through the C frame
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLREDUCE_FN(tramp) if allreduce is not None else C.cast(None, ALLREDUCE_FN)
        return check(lib().gfs_rank_run(self._h, cb, None, C.c_void_p(stream or 0)))


# ---- resident context ------------------------------------------------------------------------
class Context:
    """gfs_ctx: the graph's PathIndex mirror, positions and RNG streams resident in HBM."""

    def __init__(self, g, device=0, node_perm=None):
        self._h = C.c_void_p()
        self.graph = g
        v, self._keep = make_view(g)
        if node_perm is not None:
            node_perm = np.ascontiguousarray(node_perm, dtype=np.uint32)
        check(lib().gfs_ctx_create_with_layout(C.byref(v), C.c_int(device), _ptr(node_perm), C.byref(self._h)))
        self.dims = 0
        self.params = None
        self.cfg = None

    def close(self):
        if self._h:
            lib().gfs_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setup_1d(self, p, cfg=None, etas=None, zetas=None):
        sp = make_sgd_params(p)
        self.params, self.cfg, self.dims = p, cfg, 0
        return check(lib().gfs_ctx_setup_1d(self._h, C.byref(sp), C.byref(cfg) if cfg is not None else None,
                                            _ptr(etas), _ptr(zetas)))

    def setup_nd(self, p, cfg=None, etas=None, zetas=None):
        lp = make_layout_params(p)
        self.params, self.cfg, self.dims = p, cfg, p.dimensions
        return check(lib().gfs_ctx_setup_nd(self._h, C.byref(lp), C.byref(cfg) if cfg is not None else None,
                                            _ptr(etas), _ptr(zetas)))

    def positions_len(self):
        return int(lib().gfs_ctx_positions_len(self._h))

    def node_layout(self):
        """perm[k] = slot of dense node k in the device position vector."""
        perm = np.zeros(self.graph.n_nodes, dtype=np.uint32)
        check(lib().gfs_ctx_node_layout(self._h, _ptr(perm), C.c_uint64(perm.shape[0])))
        return perm

    def upload(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        check(lib().gfs_ctx_upload_positions(self._h, _ptr(x), x.shape[0]))

    def init_positions(self):
        """1D: the reference's initial positions, computed on the device."""
        check(lib().gfs_ctx_init_positions(self._h))

    def download(self):
        x = np.zeros(self.positions_len(), dtype=np.float64)
        check(lib().gfs_ctx_download_positions(self._h, _ptr(x), x.shape[0]))
        return x

    def positions_device(self):
        return lib().gfs_ctx_positions_device(self._h)

    def bind_positions(self, device_ptr):
        check(lib().gfs_ctx_bind_positions(self._h, C.c_void_p(device_ptr)))

    def reset_streams(self):
        check(lib().gfs_ctx_reset_streams(self._h))

    def run_iteration(self, k, stream=None):
        return check(lib().gfs_ctx_run_iteration(self._h, C.c_uint64(k), C.c_void_p(stream or 0)))

    def run_range(self, ks, stream=None):
        """Iterations ks (a fused persistent launch where possible)."""
        ks = np.ascontiguousarray(ks, dtype=np.uint64)
        return check(lib().gfs_ctx_run_range(self._h, _ptr(ks), C.c_uint64(ks.shape[0]), C.c_void_p(stream or 0)))

    def run(self, stream=None):
        return check(lib().gfs_ctx_run(self._h, C.c_void_p(stream or 0)))

    def synchronize(self, stream=None):
        check(lib().gfs_ctx_synchronize(self._h, C.c_void_p(stream or 0)))

    def stats(self):
        st = Stats()
        check(lib().gfs_ctx_stats(self._h, C.byref(st)))
        return st

    def sort_order(self):
        """Rank order of the current 1D positions, sorted on the device."""
        order = np.zeros(self.graph.n_nodes, dtype=np.uint64)
        check(lib().gfs_ctx_sort_order(self._h, _ptr(order), C.c_uint64(order.shape[0])))
        return order

    def trace(self):
        st = self.stats()
        k = int(self.cfg.trace_per_stream)
        T = int(st.n_streams)
        out = np.zeros(T * k, dtype=TERM_DTYPE)
        counts = np.zeros(T, dtype=np.uint64)
        check(lib().gfs_ctx_trace(self._h, _ptr(out), T * k, _ptr(counts), T))
        return out.reshape(T, k), counts


# ---- one-shot entry points ---------------------------------------------------------------------
def path_linear_sgd_raw(g, p, x=None, cfg=None, etas=None, zetas=None):
    """gfs_path_linear_sgd.  Returns (rc, x, stats)."""
    v, keep = make_view(g)
    sp = make_sgd_params(p)
    init = 1 if x is None else 0
    if x is None:
        x = np.zeros(g.n_nodes, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    st = Stats()
    rc = check(lib().gfs_path_linear_sgd(C.byref(v), C.byref(sp), C.byref(cfg) if cfg is not None else None,
                                         _ptr(etas), _ptr(zetas), C.c_int(init), _ptr(x), C.byref(st)))
    return rc, x, st


def path_sgd_sort_raw(g, p, x=None, cfg=None, etas=None, zetas=None):
    """gfs_path_sgd_sort.  Returns (rc, x, order, stats)."""
    v, keep = make_view(g)
    sp = make_sgd_params(p)
    init = 1 if x is None else 0
    if x is None:
        x = np.zeros(g.n_nodes, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    order = np.zeros(g.n_nodes, dtype=np.uint64)
    st = Stats()
    rc = check(lib().gfs_path_sgd_sort(C.byref(v), C.byref(sp), C.byref(cfg) if cfg is not None else None,
                                       _ptr(etas), _ptr(zetas), C.c_int(init), _ptr(x), _ptr(order), C.byref(st)))
    return rc, x, order, st


def path_linear_sgd_layout_raw(g, p, coords, cfg=None, etas=None, zetas=None):
    """gfs_path_linear_sgd_layout.  coords: float64[n_nodes*2*D] Layout order, updated copy returned."""
    v, keep = make_view(g)
    lp = make_layout_params(p)
    coords = np.ascontiguousarray(coords, dtype=np.float64).copy()
    st = Stats()
    rc = check(lib().gfs_path_linear_sgd_layout(C.byref(v), C.byref(lp), C.byref(cfg) if cfg is not None else None,
                                                _ptr(etas), _ptr(zetas), _ptr(coords), C.byref(st)))
    return rc, coords, st
