"""The N>1 path on CPU under gloo.  The plan (gfs_shard_paths / gfs_shard_quotas / gfs_shared_node_layout /
gfs_exchange_plan, host-only C++ below the ABI) and the driver logic of gfasort_amd/distributed.py (windows, the merge of
the shared slots only, idle ranks, the final completion) are the product code under test; the per-rank compute engine is
a TEST engine built on the oracle's resumable state (the HIP engine needs a GPU; tests/test_gpu_parity.py runs RankDriver)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import O, G, P, load, oracle_graph, oracle_params
from gfasort_amd.distributed import ShardedSGD


class OracleEngine:
    def __init__(self, local_graph, params, dims, quota, rank, streams_per_rank):
        self.og = oracle_graph(local_graph)
        self.op = oracle_params(params)
        self.T = streams_per_rank or 4
        n = local_graph.n_nodes * (2 * dims if dims else 1)
        self.np_x = np.zeros(n, dtype=np.float64)
        self.positions = torch.from_numpy(self.np_x)          # shares memory
        self.st = C.c_void_p()
        self.stream_base = rank * self.T
        rc = O.lib().gfo_state_create(self.og.ref, C.byref(self.op), None, None, C.c_uint64(dims),
                                      C.c_uint64(self.T), C.c_uint64(self.stream_base), C.c_uint64(max(quota, 1)),
                                      C.c_uint64(64), None, C.c_uint64(0), C.byref(self.st))
        assert rc in (0, 1)                                    # 1: nothing to do (no multi-step path in this shard)
        self.nothing_to_do = rc == 1
        self.quota = quota

    def set_positions(self, x):
        self.np_x[:] = x

    def run_iteration(self, k):
        assert not self.nothing_to_do, "an idle rank must not be asked to run"
        assert O.lib().gfo_state_run_iteration(self.st, C.c_uint64(k), self.np_x.ctypes.data_as(C.c_void_p)) == 0

    def stats(self):
        st = O.GfoStats()
        if not self.nothing_to_do:
            O.lib().gfo_state_stats(self.st, C.byref(st))
        return st


def _graph(name):
    if name == "single_step_shard":
        # 4 paths of 40 steps over a chain of 200 nodes, then 4 one-step paths: with world = 5 bin packing gives the
        # last rank the one-step paths only -> weight 0, quota 0, idle
        n = 200
        steps = [np.arange(k * 40, k * 40 + 40) for k in range(4)] + [np.array([150 + k]) for k in range(4)]
        counts = [len(s) for s in steps]
        return G.FlatGraph(node_len=(np.arange(n) % 7 + 1).astype(np.uint32), step_node=np.concatenate(steps).astype(np.uint32),
                           step_is_rev=np.zeros(sum(counts), np.uint8),
                           path_first_step=np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64),
                           node_ids=np.arange(1, n + 1, dtype=np.uint64), path_names=[f"p{k}" for k in range(8)])
    if name == "windows":
        return G.synth_windows(6000, 16, 1200, 5)
    return load(name)


def _worker(rank, world, port, name, iters, merge_every, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = _graph(name)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = iters
    r = ShardedSGD(g, p, rank, world, OracleEngine, dims=0, streams_per_rank=4, dist=dist, merge_every=merge_every,
                   sharding="lpt" if name == "single_step_shard" else "auto")
    r.set_positions(O.init_positions(oracle_graph(g)))
    r.run()
    st = r.engine.stats()
    x = r.positions_numpy()
    gathered = [torch.zeros_like(torch.from_numpy(x)) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(x))
    upd = torch.tensor([float(st.term_updates)], dtype=torch.float64)
    dist.all_reduce(upd)
    facts = torch.tensor([float(r.idle), float(r.engine.stream_base), float(r.idx.shape[0])], dtype=torch.float64)
    all_facts = [torch.zeros_like(facts) for _ in range(world)]
    dist.all_gather(all_facts, facts)
    if rank == 0:
        out.put((x, [t.numpy() for t in gathered], float(upd.item()), r.quotas, r.shards, [f.tolist() for f in all_facts]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(name, iters, world, merge_every=1):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, iters, merge_every, out)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = out.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return res


def _check_common(name, iters, world, res):
    x, gathered, total_upd, quotas, shards, facts = res
    g = _graph(name)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    for k in range(1, world):                                  # after finish() every replica is complete and identical
        assert np.array_equal(gathered[0], gathered[k]), k
    assert np.isfinite(x).all()
    # every iteration applied exactly min_term_updates updates across the ranks
    assert sum(quotas) == p.min_term_updates and total_upd == (iters + 1) * p.min_term_updates
    assert sorted(q for s in shards for q in s) == list(range(g.n_paths))
    assert [f[1] for f in facts] == [4.0 * r for r in range(world)]          # stream bases: rank * streams per rank
    assert len({f[2] for f in facts}) == 1                                    # all ranks exchange the same number of slots
    return g, p, x, quotas, facts


def test_two_ranks_gloo_delta_allreduce():
    name, iters, world = "DRB1-3123.gfa", 30, 2
    g, p, x, quotas, facts = _check_common(name, iters, world, _run(name, iters, world))
    # quality parity with the single-rank run at equal update counts (P2)
    og = oracle_graph(g)
    x0 = O.init_positions(og)
    s0 = O.stress_1d(og, x0, 20000)
    p.iter_max = iters
    x1 = x0.copy()
    O.sgd_1d(og, oracle_params(p), x1, n_streams=8)
    s_single, s_multi = O.stress_1d(og, x1, 20000), O.stress_1d(og, x, 20000)
    assert s_multi < 0.5 * s0
    assert s_multi < 1.5 * s_single + 0.05, (s0, s_single, s_multi)


def test_more_ranks_than_paths_leaves_a_rank_idle():
    """lil.gfa has 3 paths: at world 4 one rank owns nothing.  It must hold a full replica, contribute zero deltas of the
    same size as its peers to every merge, and end with the same positions (round 1 sent a (2,0) buffer and hung)."""
    name, iters, world = "lil.gfa", 20, 4
    g, p, x, quotas, facts = _check_common(name, iters, world, _run(name, iters, world))
    assert sorted(quotas) == [0, 10, 10, 10] and sum(f[0] for f in facts) == 1.0
    og = oracle_graph(g)
    assert O.stress_1d(og, x, 5000) < O.stress_1d(og, O.init_positions(og), 5000)


def res_shards(name, world):
    from gfasort_amd import hip
    g = _graph(name)
    plan = hip.ShardPlan(g, int(P.YgsParams.from_graph(g, 0, 1).path_sgd.min_term_updates), world, 2)
    return [plan.paths_of(r) for r in range(world)]


def test_a_shard_of_single_step_paths_is_idle():
    """One-step paths weigh nothing (sgd.rs:448: they never yield a term): the rank that owns only such paths gets no
    updates, the others share all of them."""
    name, iters, world = "single_step_shard", 10, 5
    g, p, x, quotas, facts = _check_common(name, iters, world, _run(name, iters, world))
    assert quotas == [41, 41, 41, 41, 0] and [f[0] for f in facts] == [0.0, 0.0, 0.0, 0.0, 1.0]
    assert res_shards(name, world)[4] == [4, 5, 6, 7]
    # nodes 160..199 are on no path at all: no rank's span covers them, and after finish() they must sit where they
    # started on every replica (sgd.rs:286-294: untouched nodes keep their cumulative-length start) — a finish() that sums
    # "what I own" with nobody owning them would zero them
    x0 = O.init_positions(oracle_graph(g))
    assert np.array_equal(x[160:], x0[160:]) and (x0[160:] > 0).all()


def test_two_ranks_leave_unvisited_nodes_where_they_started():
    name, iters, world = "single_step_shard", 6, 2
    g, p, x, quotas, facts = _check_common(name, iters, world, _run(name, iters, world))
    x0 = O.init_positions(oracle_graph(g))
    assert np.array_equal(x[160:], x0[160:]) and (x0[160:] > 0).all()


def test_eight_ranks_windows_exchange_only_the_overlaps():
    """8 ranks on a window graph, merge every 2 iterations: sharding, quotas, stream bases; only the slots where
    neighbouring ranks' spans overlap are exchanged, yet after finish() all replicas are complete and equal, and the
    sort is as good as the single-rank run's."""
    name, iters, world = "windows", 40, 8
    g, p, x, quotas, facts = _check_common(name, iters, world, _run(name, iters, world, merge_every=2))
    assert 0 < facts[0][2] < g.n_nodes                              # a strict subset of the slots travels per window
    assert max(quotas) - min(quotas) <= 0.11 * max(quotas)
    og = oracle_graph(g)
    x0 = O.init_positions(og)
    p.iter_max = iters
    x1 = x0.copy()
    O.sgd_1d(og, oracle_params(p), x1, n_streams=8)
    s0, s1, s8 = O.stress_1d(og, x0, 20000), O.stress_1d(og, x1, 20000), O.stress_1d(og, x, 20000)
    assert s8 < 0.05 * s0 and s8 < 3.0 * s1 + 1e-3, (s0, s1, s8)


def test_world_size_1_takes_no_collective_and_equals_plain_run():
    g = load("lil.gfa")
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    r = ShardedSGD(g, p, 0, 1, OracleEngine, dims=0, streams_per_rank=4, dist=None)
    og = oracle_graph(g)
    r.set_positions(O.init_positions(og))
    r.run()
    x_ref = O.init_positions(og)
    O.sgd_1d(og, oracle_params(p), x_ref, n_streams=4)
    assert np.array_equal(r.positions_numpy(), x_ref)
    assert r.x_prev is None
