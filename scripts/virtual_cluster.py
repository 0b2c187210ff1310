"""Quality of the multi-GPU merge with the REAL product path on ONE GPU: R rank threads, each a gfs_rank (RankDriver) on
device 0, the collective a barrier-based in-process sum of the ranks' torch tensors.  Throughput means nothing here; what is
measured is what 8 ranks' sharded sampling + per-window merge does to the layout: relative error per octave of path distance
(distance 1 and 2-3 over all pairs), sampled stress, against the single-GPU run, at merge_every 1 and 4.
    python scripts/virtual_cluster.py [world = 8] [rules = touch[,anneal,...]]   (GFS_DBG_MERGE_ETA_FACTOR scales rule 3's switch)"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
from gfasort_amd import graph as G, params as P, hip, quality as Q   # noqa: E402
from gfasort_amd.distributed import RankDriver   # noqa: E402


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


def run_cluster(g, p, world, merge_every=1, sharding="auto", merge="touch"):
    dist = FakeDist(world)
    out = [None] * world
    errs = []

    def work(rank):
        try:
            dist.bind(rank)
            torch.cuda.set_device(0)
            r = RankDriver(g, p, rank, world, dims=0, device_index=0, dist=dist, merge_every=merge_every, sharding=sharding, merge=merge)
            r.set_positions(None)
            r.run()
            torch.cuda.synchronize()
            out[rank] = (r.positions_numpy(), int(r.info.shared_slots), int(r.info.quota))
            r.close()
        except Exception as e:          # noqa
            import traceback
            traceback.print_exc()
            errs.append(e)
            dist.bar.abort()

    th = [threading.Thread(target=work, args=(k,)) for k in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    assert all(np.array_equal(out[0][0], o[0]) for o in out), "replicas differ"
    return out[0][0], out[0][1], [o[2] for o in out]


def profile(g, x):
    _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
    return rms[:10]


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cases = [("bubbles 525k nodes, 24 paths", G.synth_bubbles(400_000, 24, 6), 100, False),
             ("windows C3 1M nodes, 64 paths", G.synth_windows(1_000_000, 64, 156_250, 2), 200, True)]
    for name, g, iters, chain in cases:
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = iters
        rc, x1, st = hip.path_linear_sgd_raw(g, p)
        p1 = profile(g, x1)
        s1 = Q.sampled_stress(g, x1, 0, 2_000_000)
        extra = f" inversions {Q.inversions_vs_chain(g.node_ids[hip.sort_order(x1).astype(np.int64)].astype(np.int64))}" if chain else ""
        print(f"{name}, -p Y --iter-max {iters}\n  single GPU (B {st.bundle}): stress {s1:.4e}{extra}  profile " + " ".join(f"{v:.4f}" for v in p1), flush=True)
        rules = sys.argv[2].split(",") if len(sys.argv) > 2 else ["touch"]
        for merge, merge_every in [(m, e) for m in rules for e in (1, 4)]:
            x, shared, quotas = run_cluster(g, p, world, merge_every=merge_every, merge=merge)
            pw = profile(g, x)
            s = Q.sampled_stress(g, x, 0, 2_000_000)
            extra = f" inversions {Q.inversions_vs_chain(g.node_ids[hip.sort_order(x).astype(np.int64)].astype(np.int64))}" if chain else ""
            print(f"  {world} ranks, rule {merge}, merge every {merge_every}: shared slots {shared} of {g.n_nodes}, quotas {min(quotas)}..{max(quotas)}; stress {s:.4e} "
                  f"(x{s / s1:.2f}){extra}  profile " + " ".join(f"{v:.4f}" for v in pw) + "  ratio " + " ".join(f"{a / b:.2f}" for a, b in zip(pw, p1)),
                  flush=True)


if __name__ == "__main__":
    main()
