"""Quality of the multi-GPU merge rules with the REAL kernels: R rank threads share one GPU, the
collective is a barrier-based in-process all-reduce.  (Throughput numbers here mean nothing.)"""
import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gfasort_amd import graph as G, params as P, hip
from gfasort_amd.distributed import ShardedSGD, hip_engine_factory
from oracle import oracle as O


class FakeDist:
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.result = None
        self.local = threading.local()

    def bind(self, rank):
        self.local.rank = rank

    def all_reduce(self, t):
        r = self.local.rank
        torch.cuda.synchronize()
        self.slots[r] = t
        self.bar.wait()
        if r == 0:
            acc = self.slots[0].clone()
            for k in range(1, self.world):
                acc += self.slots[k]
            self.result = acc
            torch.cuda.synchronize()
        self.bar.wait()
        t.copy_(self.result)
        torch.cuda.synchronize()
        self.bar.wait()


def run_cluster(g, p, world, merge="touch", merge_every=1, sharding="auto", flags=0):
    dist = FakeDist(world)
    out = [None] * world
    errs = []

    def work(rank):
        try:
            dist.bind(rank)
            r = ShardedSGD(g, p, rank, world, hip_engine_factory(device_index=0, flags=flags), dims=0,
                           merge=merge, dist=dist, merge_every=merge_every, sharding=sharding)
            r.set_positions(hip.init_positions(g))
            r.run()
            torch.cuda.synchronize()
            out[rank] = r.positions_numpy()
            r.engine.close()
        except Exception as e:          # noqa
            errs.append(e)
            dist.bar.abort()

    th = [threading.Thread(target=work, args=(k,)) for k in range(world)]
    for t in th: t.start()
    for t in th: t.join()
    if errs:
        raise errs[0]
    assert all(np.array_equal(out[0], o) for o in out)
    return out[0]


def quality(g, x, chain):
    og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
    s = O.stress_1d(og, x, 200000)
    if not chain:
        return f"stress {s:.4e}"
    ids = g.node_ids[np.argsort(x, kind="stable")].astype(np.int64)
    fwd = ids if ids[0] < ids[-1] else ids[::-1]
    return f"stress {s:.3e} inversions {int((np.diff(fwd) < 0).sum())}"


def main():
    cases = [("C3", G.synth_windows(1_000_000, 64, 156_250, 2), 200, True),
             ("bub400k", G.synth_bubbles(400000, 24, 6), 100, False)]
    for name, g, iters, chain in cases:
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = iters
        rc, x1, st = hip.path_linear_sgd_raw(g, p)
        print(f"{name} single GPU: {quality(g, x1, chain)}", flush=True)
        for world in (8,):
            for sharding in ("contiguous", "lpt"):
                for merge_every in (1, 4, 16):
                    for merge in ("touch",):
                        x = run_cluster(g, p, world, merge=merge, merge_every=merge_every, sharding=sharding)
                        print(f"{name} world {world} sharding {sharding:10s} merge {merge} every {merge_every:2d}: {quality(g, x, chain)}", flush=True)

if __name__ == "__main__":
    main()
