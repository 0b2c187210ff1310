"""The N>1 path on CPU: world_size 2, gloo.  The sharding / delta all-reduce driver
(gfasort_amd/distributed.py ShardedSGD) is the product code under test; the per-rank compute
engine is a TEST engine built on the oracle's resumable state (the HIP engine needs a GPU)."""
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
        rc = O.lib().gfo_state_create(self.og.ref, C.byref(self.op), None, None, C.c_uint64(dims),
                                      C.c_uint64(self.T), C.c_uint64(rank * self.T), C.c_uint64(max(quota, 1)),
                                      C.c_uint64(64), None, C.c_uint64(0), C.byref(self.st))
        assert rc == 0
        self.quota = quota

    def set_positions(self, x):
        self.np_x[:] = x

    def run_iteration(self, k):
        assert O.lib().gfo_state_run_iteration(self.st, C.c_uint64(k), self.np_x.ctypes.data_as(C.c_void_p)) == 0

    def stats(self):
        st = O.GfoStats()
        O.lib().gfo_state_stats(self.st, C.byref(st))
        return st


def _worker(rank, world, port, name, iters, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load(name)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = iters
    r = ShardedSGD(g, p, rank, world, OracleEngine, dims=0, streams_per_rank=4, dist=dist)
    r.set_positions(O.init_positions(oracle_graph(g)))
    r.run()
    st = r.engine.stats()
    x = r.positions_numpy()
    gathered = [torch.zeros_like(torch.from_numpy(x)) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(x))
    upd = torch.tensor([float(st.term_updates)], dtype=torch.float64)
    dist.all_reduce(upd)
    if rank == 0:
        out.put((x, [t.numpy() for t in gathered], float(upd.item()), r.quotas, r.shards))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_gloo_delta_allreduce():
    name, iters, world = "DRB1-3123.gfa", 30, 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, iters, out)) for r in range(world)]
    for pr in procs:
        pr.start()
    x, gathered, total_upd, quotas, shards = out.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    g = load(name)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    # replicas agree bit for bit after the merge
    assert np.array_equal(gathered[0], gathered[1])
    # every iteration applied exactly min_term_updates updates across the ranks
    assert sum(quotas) == p.min_term_updates and total_upd == (iters + 1) * p.min_term_updates
    assert sorted(q for s in shards for q in s) == list(range(g.n_paths))
    # quality parity with the single-rank run at equal update counts (P2)
    og = oracle_graph(g)
    x0 = O.init_positions(og)
    s0 = O.stress_1d(og, x0, 20000)
    p.iter_max = iters
    x1 = x0.copy()
    O.sgd_1d(og, oracle_params(p), x1, n_streams=8)
    s_single, s_multi = O.stress_1d(og, x1, 20000), O.stress_1d(og, x, 20000)
    assert np.isfinite(x).all()
    assert s_multi < 0.5 * s0
    assert s_multi < 1.5 * s_single + 0.05, (s0, s_single, s_multi)


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
