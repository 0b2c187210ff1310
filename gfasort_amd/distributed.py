"""Multi-GPU SGD: paths sharded over ranks, one process per GPU, replicas merged after every window of
iterations with ONE all-reduce (RCCL over xGMI when the tensors live on GPUs).

The reference is single-process (Hogwild threads on one shared vector, sgd.rs:413-593); this is the MI355X-native
scale-out of it.  The logic lives BELOW the C ABI (gfasort_amd/csrc/multi.hip, include/gfasort_hip.h
"multi-device runs"): gfs_shard_paths / gfs_shard_quotas / gfs_shared_node_layout / gfs_exchange_plan plan the run on
the host, gfs_rank is one rank on one device.  This module is the binding that supplies the collective:

  * `RankDriver`  — the product path: a gfs_rank plus torch.distributed (backend "nccl" = RCCL).  The exchange buffer is
                    a torch tensor bound into the rank, so the all-reduce runs on it in place.
  * `ShardedSGD`  — the same plan and the same merge written with torch ops, around an engine that keeps its positions
                    in a torch tensor in ABI order.  Tests inject a CPU engine built on the oracle to exercise
                    sharding, quotas, stream bases, idle ranks and the merge under gloo; it is not a product path.

How a window is merged (both classes):
  * rank r owns a subset of the paths and samples step a only from them; its share of an iteration's term updates is
    proportional to its share of the steps of multi-step paths, so the global sampling distribution stays uniform;
  * only the slots that two or more ranks' paths can move are exchanged (all ranks share one node layout; a rank's
    paths touch one span of it): buf = [delta, touched], delta_r = x_r - x_prev, touched_r = delta_r != 0, and
        x <- x_prev + sum_r delta_r / max(1, sum_r touched_r)                            (merge="touch")
    and the product's default, merge="anneal" (RankDriver / gfs_rank merge rule 0), lets that divisor fall from the number of
    ranks that moved the node to 1 as the learning rate falls below the scale of the shortest terms:
        x <- x_prev + sum_r delta_r / max(1, sum_r touched_r * min(1, window length * eta / mean node length))
    (the mean of c full corrections early, the sum of c sets of small steps late — averaging those threw away 7/8 of a
    window's work at 8 ranks exactly where the layout is finished: relative error at path distance 1 1.47x the single-GPU
    run's under "touch", 1.07x under "anneal", profiles/r03/virtual_cluster.log).
    A node moved by one rank only receives that rank's full move; a node moved by c ranks receives the mean of the c
    proposals.  Plain summation (merge="sum") applies c full corrections of the same error and diverges for c >= 3
    while the learning rate is still clamped at mu = 1 (measured: stress 1e8 on DRB1 at 4 ranks); plain averaging over
    all ranks (merge="mean") is stable but under-applies moves of nodes that few ranks touched;
  * slots inside one rank's span only never travel until `finish()`: one full-length f64 all-reduce of "what I own";
  * a rank whose shard has no multi-step path (or whose quota is 0) is idle: it runs no kernel and contributes zero
    deltas of the same size as everybody else.
World size 1 takes none of this path: no collective, no extra kernels.
"""
from typing import Callable, List

import numpy as np

from .graph import FlatGraph

SHARDING = {"auto": 0, "contiguous": 1, "lpt": 2}


def shard_quotas(total_updates: int, shard_steps: List[int]) -> List[int]:
    """gfs_shard_quotas: an iteration's term updates split proportionally to step counts (largest remainder)."""
    from . import hip
    import ctypes as C
    steps = np.ascontiguousarray(shard_steps, dtype=np.uint64)
    out = np.zeros(len(shard_steps), dtype=np.uint64)
    hip.check(hip.lib().gfs_shard_quotas(C.c_uint64(int(total_updates)), hip._ptr(steps), len(shard_steps), hip._ptr(out)))
    return [int(v) for v in out]


def path_order_layout(g: FlatGraph) -> np.ndarray:
    """gfs_shared_node_layout: perm[k] = rank of dense node k in first-visit path order with branches placed where they branch off (unvisited nodes last) — the
    rule libgfasort_hip applies by default, here on the WHOLE graph so that every rank stores its replica alike."""
    from . import hip
    import ctypes as C
    v, keep = hip.make_view(g)
    perm = np.zeros(max(g.n_nodes, 1), dtype=np.uint32)
    hip.check(hip.lib().gfs_shared_node_layout(C.byref(v), hip._ptr(perm)))
    return perm[:g.n_nodes]


def subgraph(g: FlatGraph, path_ids: List[int]) -> FlatGraph:
    """The graph restricted to some paths; all nodes are kept."""
    first = g.path_first_step.astype(np.int64)
    segs = [np.arange(first[p], first[p + 1]) for p in path_ids]
    idx = np.concatenate(segs) if segs else np.zeros(0, dtype=np.int64)
    counts = np.array([first[p + 1] - first[p] for p in path_ids], dtype=np.int64)
    return FlatGraph(
        node_len=g.node_len,
        step_node=g.step_node[idx],
        step_is_rev=g.step_is_rev[idx],
        path_first_step=np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64),
        node_ids=g.node_ids,
        path_names=[g.path_names[p] for p in path_ids] if g.path_names else [],
    )


class _Windows:
    """Iterations -> merge windows: a window ends after every `merge_every` iterations and after the last one."""

    def __init__(self, iter_max: int, merge_every: int, merging: bool):
        self.iter_max, self.every, self.merging = int(iter_max), max(1, int(merge_every)), merging

    def due(self, k: int) -> bool:
        return self.merging and ((k + 1) % self.every == 0 or k == self.iter_max)

    def split(self, ks):
        seg = []
        for k in ks:
            seg.append(int(k))
            if self.due(int(k)):
                yield seg, True
                seg = []
        if seg:
            yield seg, False


class RankDriver:
    """One rank of the product: gfs_rank on one MI355X, the collective from torch.distributed.

    world_for_plan lets a test on a one-GPU box plan for more ranks than the process group has (a one-rank "nccl"
    group is the only way to run RCCL there): the kernels and the collective run, the peers' moves are simply absent."""

    def __init__(self, graph: FlatGraph, params, rank: int, world: int, dims: int = 0, device_index: int = 0,
                 streams_per_rank: int = 0, flags: int = 0, block_size: int = 0, merge: str = "anneal", dist=None,
                 merge_every: int = 1, sharding: str = "auto", whole_vector: bool = False, payload_f64: bool = False,
                 profile: bool = False):
        import torch
        from . import hip
        self._torch, self.hip = torch, hip
        self.rank, self.world, self.dist = rank, world, dist
        self.params, self.dims = params, dims
        self.device = torch.device("cuda", device_index)
        launch = hip.make_config(n_streams=streams_per_rank, flags=flags, block_size=block_size)
        self.r = hip.Rank(graph, params, dims, rank, world, device=device_index, sharding=SHARDING[sharding],
                          merge_every=merge_every, merge_rule=merge, payload_f64=payload_f64, whole_vector=whole_vector,
                          launch=launch)
        self.info = self.r.info()
        self.windows = _Windows(params.iter_max, merge_every, world > 1)
        self.buf = None
        if world > 1 and self.info.exchange_count:
            self.buf = torch.zeros(int(self.info.exchange_count), dtype=torch.float64 if payload_f64 else torch.float32,
                                   device=self.device)
            self.r.bind_exchange_buffer(self.buf.data_ptr())
        self.full = None
        self.profile = profile
        self.timing = {"compute_ms": 0.0, "exchange_ms": 0.0, "windows": 0, "exchange_bytes_per_window":
                       int(self.info.exchange_count) * (8 if payload_f64 else 4)}
        self._events = []

    def _stream(self):
        return self._torch.cuda.current_stream(self.device).cuda_stream

    def set_positions(self, x=None):
        self.r.set_positions(x)

    def reset_streams(self):
        self.r.reset_streams()

    def _ev(self):
        e = self._torch.cuda.Event(enable_timing=True)
        e.record(self._torch.cuda.current_stream(self.device))
        return e

    def run_range(self, ks):
        st = self._stream()
        for seg, merge in self.windows.split(ks):
            e0 = self._ev() if self.profile else None
            self.r.window_begin(seg, st)
            if merge:
                e1 = self._ev() if self.profile else None
                if self.buf is not None:
                    self.dist.all_reduce(self.buf)                  # RCCL over xGMI, on torch's current stream
                self.r.window_end(st)
                if self.profile:
                    self._events.append((e0, e1, self._ev()))
            elif self.profile:
                self._events.append((e0, self._ev(), None))

    def run_iteration(self, k):
        self.run_range([k])

    def run(self):
        self.run_range(range(int(self.params.iter_max) + 1))
        self.finish()

    def finish(self):
        """Complete the replica: slots only one rank moves are current on that rank alone until now."""
        if self.world < 2:
            return
        torch = self._torch
        if self.full is None:
            self.full = torch.zeros(int(self.info.positions_len), dtype=torch.float64, device=self.device)
        st = self._stream()
        self.r.finish_begin(self.full.data_ptr(), st)
        self.dist.all_reduce(self.full)
        self.r.finish_end(self.full.data_ptr(), st)

    def collect_timing(self):
        """Sum the recorded events (synchronises): compute (kernels of the windows + prepare) vs exchange (all-reduce + apply)."""
        self._torch.cuda.synchronize(self.device)
        for e0, e1, e2 in self._events:
            self.timing["compute_ms"] += e0.elapsed_time(e1)
            if e2 is not None:
                self.timing["exchange_ms"] += e1.elapsed_time(e2)
                self.timing["windows"] += 1
        self._events = []
        return dict(self.timing)

    def positions_numpy(self) -> np.ndarray:
        return self.r.get_positions()

    def stats(self):
        return self.r.ctx_stats()

    def close(self):
        self.r.close()


class ShardedSGD:
    """The same run around a caller-supplied engine (tests: a CPU engine on the oracle, gloo).
    `engine_factory(local_graph, params, dims, quota, rank, streams_per_rank)` returns an engine with
    .positions (torch tensor, float64, ABI order: x[dense node] or Layout.coords), .set_positions(x),
    .run_iteration(k) and .stats()."""

    def __init__(self, graph: FlatGraph, params, rank: int, world: int, engine_factory: Callable,
                 dims: int = 0, streams_per_rank: int = 0, merge: str = "anneal", dist=None,
                 merge_every: int = 1, sharding: str = "auto", whole_vector: bool = False, payload_f64: bool = False):
        from . import hip
        self.rank, self.world, self.merge, self.dist = rank, world, merge, dist
        self.params, self.dims = params, dims
        # merge="anneal" (gfs_rank merge rule 0): the schedule and the scale of a short-range term
        self.etas = hip.sgd_schedule(params)
        self.eta_sum = max(1.0, float(graph.node_len.astype(np.float64).mean())) if graph.n_nodes else 1.0
        self.plan = hip.ShardPlan(graph, int(params.min_term_updates), world, SHARDING[sharding], whole_vector)
        self.shards = [self.plan.paths_of(r) for r in range(world)]
        self.quotas = [int(q) for q in self.plan.quotas]
        local = graph if world == 1 else subgraph(graph, self.shards[rank])
        self.local_graph = local
        local.shared_node_layout = self.plan.perm if world > 1 else None
        self.idle = world > 1 and self.quotas[rank] == 0
        self.engine = engine_factory(local, params, dims, self.quotas[rank] if world > 1 else int(params.min_term_updates),
                                     rank, streams_per_rank)
        self.windows = _Windows(params.iter_max, merge_every, world > 1)
        self.payload_f64 = payload_f64
        self.x_prev = None
        if world > 1:
            import torch
            self._torch = torch
            n = graph.n_nodes
            width = 2 * dims if dims else 1
            slot_shared = np.zeros(n + 1, dtype=np.int64)
            for lo, hi in self.plan.shared:
                slot_shared[lo] += 1
                slot_shared[hi] -= 1
            in_shared = np.cumsum(slot_shared[:-1]) > 0
            nodes = np.flatnonzero(in_shared[self.plan.perm.astype(np.int64)])
            self.idx = torch.from_numpy((nodes[:, None] * width + np.arange(width)[None, :]).reshape(-1).astype(np.int64))
            owner = np.full(n, -1, dtype=np.int64)
            for lo, hi, r in self.plan.owned:
                owner[lo:hi] = r
            mine = owner[self.plan.perm.astype(np.int64)] == rank
            self.mine = torch.from_numpy(np.repeat(mine, width))
            self.x_prev = self.engine.positions[self.idx].clone()

    def set_positions(self, x: np.ndarray):
        self.engine.set_positions(x)
        if self.world > 1:
            self.x_prev = self.engine.positions[self.idx].clone()

    def run_iteration(self, k: int):
        self.run_range([k])

    def run_range(self, ks):
        for seg, merge in self.windows.split(ks):
            if not self.idle:
                for k in seg:
                    self.engine.run_iteration(k)
            if merge:
                self._merge(seg)

    def _merge(self, seg=()):
        torch = self._torch
        x = self.engine.positions
        dt = torch.float64 if self.payload_f64 else torch.float32
        d = x[self.idx] - self.x_prev                                # this rank's moves of the shared slots
        buf = torch.stack([d.to(dt), (d != 0).to(dt)])
        self.dist.all_reduce(buf)
        if self.merge == "touch":
            div = buf[1].clamp(min=1.0).to(x.dtype)
        elif self.merge == "anneal":
            cscale = min(1.0, len(seg) * float(self.etas[seg[-1]]) / self.eta_sum) if len(seg) else 1.0
            div = (buf[1].to(x.dtype) * cscale).clamp(min=1.0)
        else:
            div = 1.0 if self.merge == "sum" else float(self.world)
        self.x_prev = self.x_prev + buf[0].to(x.dtype) / div
        x[self.idx] = self.x_prev

    def finish(self):
        if self.world < 2:
            return
        torch = self._torch
        x = self.engine.positions
        full = torch.where(self.mine, x, torch.zeros_like(x))
        self.dist.all_reduce(full)
        x.copy_(full)
        self.x_prev = x[self.idx].clone()

    def run(self):
        self.run_range(range(int(self.params.iter_max) + 1))
        self.finish()

    def positions_numpy(self) -> np.ndarray:
        """Positions in the ABI's dense-index order."""
        if hasattr(self.engine, "get_positions"):
            return self.engine.get_positions()
        return self.engine.positions.detach().cpu().numpy().copy()


class HipEngine:
    """One gfs_ctx on one MI355X behind the engine interface (single-GPU legs of bench.py; a test engine for
    ShardedSGD is NOT what this is for: N > 1 on GPUs goes through RankDriver)."""

    def __init__(self, local_graph, params, dims, quota, rank, streams_per_rank, device_index=0,
                 flags=0, block_size=0):
        import torch
        from . import hip
        self._torch = torch
        self.hip = hip
        self.device = torch.device("cuda", device_index)
        self.ctx = hip.Context(local_graph, device=device_index,
                               node_perm=getattr(local_graph, "shared_node_layout", None))
        self.dims = dims
        cfg = hip.make_config(n_streams=streams_per_rank, stream_base=rank * (streams_per_rank or (1 << 20)),
                              term_updates_per_iteration=quota, flags=flags, block_size=block_size)
        self.cfg = cfg
        rc = self.ctx.setup_nd(params, cfg) if dims else self.ctx.setup_1d(params, cfg)
        self.nothing_to_do = rc == hip.NOTHING_TO_DO

    def set_positions(self, x):
        self.ctx.upload(np.ascontiguousarray(x, dtype=np.float64))

    def get_positions(self):
        return self.ctx.download()

    def reset_streams(self):
        self.ctx.reset_streams()

    def run_iteration(self, k):
        stream = self._torch.cuda.current_stream(self.device).cuda_stream
        return self.ctx.run_iteration(k, stream)

    def run_range(self, ks):
        stream = self._torch.cuda.current_stream(self.device).cuda_stream
        return self.ctx.run_range(ks, stream)

    def stats(self):
        return self.ctx.stats()

    def close(self):
        self.ctx.close()
