"""Multi-GPU SGD: paths sharded over ranks, positions replicated, one all-reduce of position
deltas per iteration (RCCL over xGMI when the tensors live on GPUs).

The reference is single-process (Hogwild threads on one shared vector, sgd.rs:413-593); this is
the MI355X-native scale-out of it:
  * rank r owns a subset of the paths (longest-first bin packing on step counts) and samples
    step a only from its own steps; its share of an iteration's term updates is proportional
    to its share of the steps, so the global sampling distribution stays uniform over steps;
  * every rank runs its batch on its own replica of the positions, then the replicas are merged
    with ONE all-reduce of [delta, touched] in f32 (delta_r = x_r - x_prev, touched_r = delta_r != 0):
        x <- x_prev + sum_r delta_r / max(1, sum_r touched_r)            (merge="touch", default)
    A node moved by one rank only receives that rank's full move; a node moved by c ranks
    receives the mean of the c proposals.  Plain summation (merge="sum") applies c full
    corrections of the same error and diverges for c >= 3 while the learning rate is still
    clamped at mu = 1 (measured: stress 1e8 on DRB1 at 4 ranks); plain averaging over all
    ranks (merge="mean") is stable but under-applies moves of nodes that few ranks touched.
  * one process per GPU; torch.distributed supplies the collective (backend "nccl" = RCCL).
World size 1 takes none of this path: no collective, no extra kernels.
"""
from dataclasses import replace
from typing import Callable, List, Optional

import numpy as np

from .graph import FlatGraph


def shard_paths_contiguous(step_counts: np.ndarray, world: int) -> List[List[int]]:
    """Consecutive blocks of paths with (nearly) equal step totals: rank r gets the paths whose
    cumulative-step midpoint falls into the r-th 1/world of the total."""
    counts = np.asarray(step_counts, dtype=np.float64)
    total = float(counts.sum())
    shards: List[List[int]] = [[] for _ in range(world)]
    if total <= 0:
        return shards
    mid = np.cumsum(counts) - counts / 2.0
    owner = np.minimum((mid * world / total).astype(np.int64), world - 1)
    for p, r in enumerate(owner.tolist()):
        shards[r].append(p)
    return shards


def shard_imbalance(step_counts: np.ndarray, shards: List[List[int]]) -> float:
    loads = [float(sum(int(step_counts[p]) for p in s)) for s in shards]
    mean = sum(loads) / max(len(loads), 1)
    return (max(loads) / mean - 1.0) if mean > 0 else 0.0


def shard_paths(step_counts: np.ndarray, world: int) -> List[List[int]]:
    """Longest-processing-time-first bin packing of paths onto ranks (deterministic)."""
    order = sorted(range(len(step_counts)), key=lambda p: (-int(step_counts[p]), p))
    loads = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for p in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(p)
        loads[r] += int(step_counts[p])
    for s in shards:
        s.sort()
    return shards


def shard_quotas(total_updates: int, shard_steps: List[int]) -> List[int]:
    """Split an iteration's term updates proportionally to step counts (largest remainder),
    summing exactly to total_updates."""
    S = sum(shard_steps)
    if S == 0:
        return [0] * len(shard_steps)
    base = [total_updates * s // S for s in shard_steps]
    rem = total_updates - sum(base)
    frac = sorted(range(len(shard_steps)), key=lambda k: (-(total_updates * shard_steps[k] % S), k))
    for k in frac[:rem]:
        base[k] += 1
    return base


def path_order_layout(g: FlatGraph) -> np.ndarray:
    """perm[k] = rank of dense node k in first-visit path order (unvisited nodes last): the same
    rule libgfasort_hip applies by default, computed on the WHOLE graph so that every rank of a
    multi-GPU run stores its position replica in the same order."""
    n = g.n_nodes
    valid = g.step_node[g.step_node != 0xFFFFFFFF].astype(np.int64)
    perm = np.full(n, -1, dtype=np.int64)
    if valid.size:
        uniq, first = np.unique(valid, return_index=True)
        visited = uniq[np.argsort(first, kind="stable")]
        perm[visited] = np.arange(visited.size)
        nxt = visited.size
    else:
        nxt = 0
    rest = np.flatnonzero(perm < 0)
    perm[rest] = nxt + np.arange(rest.size)
    return perm.astype(np.uint32)


def subgraph(g: FlatGraph, path_ids: List[int]) -> FlatGraph:
    """The graph restricted to some paths; all nodes are kept (positions are replicated)."""
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


class ShardedSGD:
    """Driver for one rank.  `engine_factory(local_graph, params, dims, quota, stream_base)`
    returns an engine with: .positions (a torch tensor on the compute device, float64, the
    engine's live position buffer), .run_iteration(k), .stats().  The product engine is
    HipEngine below; tests inject a CPU engine to exercise this logic under gloo."""

    def __init__(self, graph: FlatGraph, params, rank: int, world: int, engine_factory: Callable,
                 dims: int = 0, streams_per_rank: int = 0, merge: str = "touch", dist=None,
                 merge_every: int = 1, sharding: str = "auto", force_merge: bool = False):
        self.rank, self.world, self.merge = rank, world, merge
        # force_merge: run the merge (kernels + collective) even with one rank — a 1-rank RCCL group is the
        # only way to exercise the real collective path on a one-GPU machine (tests)
        self.merging = world > 1 or (force_merge and dist is not None)
        self.dist = dist
        self.merge_every = max(1, int(merge_every))
        counts = graph.path_step_counts()
        # consecutive blocks keep a rank's paths (and so the nodes it moves) together when the paths
        # of the input are ordered along the graph; fall back to LPT packing when that is unbalanced
        contiguous = shard_paths_contiguous(counts, world)
        if sharding == "contiguous" or (sharding == "auto" and shard_imbalance(counts, contiguous) <= 0.10):
            self.shards = contiguous
        else:
            self.shards = shard_paths(counts, world)
        steps = [int(sum(int(counts[p]) for p in s)) for s in self.shards]
        self.quotas = shard_quotas(int(params.min_term_updates), steps)
        local = graph if world == 1 else subgraph(graph, self.shards[rank])
        self.local_graph = local
        self.params = params
        self.stream_stride = streams_per_rank
        # one node layout for all ranks (each rank only sees its own paths)
        local.shared_node_layout = path_order_layout(graph) if world > 1 else None
        self.engine = engine_factory(local, params, dims, self.quotas[rank], rank, streams_per_rank)
        self.x_prev = None
        self._buf = None
        if self.merging:
            import torch
            self._torch = torch
            self.x_prev = torch.empty_like(self.engine.positions)

    def set_positions(self, x: np.ndarray):
        self.engine.set_positions(x)
        if self.merging:
            self.x_prev.copy_(self.engine.positions)

    def _merge_due(self, k: int) -> bool:
        return self.merging and ((k + 1) % self.merge_every == 0 or k == int(self.params.iter_max))

    def run_iteration(self, k: int):
        self.engine.run_iteration(k)
        self._merge_if_due(k)

    def run_range(self, ks):
        """Iterations ks in order; the iterations between two merges go to the engine as one range
        (one fused persistent launch on the HIP engine)."""
        seg = []
        for k in ks:
            seg.append(int(k))
            if self._merge_due(int(k)):
                self._run_segment(seg)
                self._merge_if_due(int(k))
                seg = []
        if seg:
            self._run_segment(seg)

    def _run_segment(self, seg):
        if hasattr(self.engine, "run_range"):
            self.engine.run_range(seg)
        else:
            for k in seg:
                self.engine.run_iteration(k)

    def _merge_if_due(self, k: int):
        if self._merge_due(k):
            torch = self._torch
            x = self.engine.positions
            n = x.shape[0]
            if self._buf is None:
                self._buf = torch.empty((2, n), dtype=torch.float32, device=x.device)   # f32 on the wire
            buf = self._buf
            divide = {"touch": 0.0, "sum": 1.0, "mean": float(self.world)}[self.merge]
            if x.is_cuda and hasattr(self.engine, "hip"):
                # fused HIP kernels on the engine's stream around the one collective
                st = torch.cuda.current_stream(x.device).cuda_stream
                self.engine.hip.merge_prepare(x.data_ptr(), self.x_prev.data_ptr(), buf.data_ptr(), n, st)
                self.dist.all_reduce(buf)                 # RCCL over xGMI
                self.engine.hip.merge_apply(x.data_ptr(), self.x_prev.data_ptr(), buf.data_ptr(), n, divide, st)
            else:
                buf[0] = (x - self.x_prev).to(torch.float32)      # this rank's batch
                buf[1] = (buf[0] != 0).to(torch.float32)
                self.dist.all_reduce(buf)
                div = buf[1].clamp(min=1.0).to(x.dtype) if divide == 0.0 else divide
                self.x_prev += buf[0].to(x.dtype) / div
                x.copy_(self.x_prev)

    def run(self):
        self.run_range(range(int(self.params.iter_max) + 1))

    def positions_numpy(self) -> np.ndarray:
        """Positions in the ABI's dense-index order."""
        if hasattr(self.engine, "get_positions"):
            return self.engine.get_positions()
        return self.engine.positions.detach().cpu().numpy().copy()


class HipEngine:
    """The product engine: gfs_ctx on one MI355X, positions in a torch CUDA tensor bound into
    the context so torch.distributed can all-reduce them in place."""

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
        n = self.ctx.positions_len()
        self.positions = torch.zeros(max(n, 1), dtype=torch.float64, device=self.device)[:n]
        if n:
            self.ctx.bind_positions(self.positions.data_ptr())

    def set_positions(self, x):
        # through the ABI: it applies the context's internal node layout
        self.ctx.upload(np.ascontiguousarray(x, dtype=np.float64))
        self._torch.cuda.synchronize(self.device)

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


def hip_engine_factory(device_index=0, flags=0, block_size=0):
    def make(local_graph, params, dims, quota, rank, streams_per_rank):
        return HipEngine(local_graph, params, dims, quota, rank, streams_per_rank,
                         device_index=device_index, flags=flags, block_size=block_size)
    return make
