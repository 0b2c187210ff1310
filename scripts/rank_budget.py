"""What ONE rank of an 8-GPU run of BASELINE's configs[4] (C5: 10M nodes, 1024 paths, 1e8 steps) spends per iteration on
everything that is not the collective, measured on the one GPU a box has: rank 0 of a plan for 8 ranks (1/8 of the paths and
of each iteration's updates), merge at every iteration, the all-reduce replaced by a no-op.  HIP events around
[kernel of the window + pack] and [apply], and the host's wall clock per iteration.
    python scripts/rank_budget.py [world = 8]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
from gfasort_amd import graph as G, params as P, hip   # noqa: E402
from gfasort_amd.distributed import RankDriver   # noqa: E402


class NoDist:
    def all_reduce(self, t):
        return None


world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = G.synth_windows(10_000_000, 1024, 97_656, 3)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
p.iter_max = 100
torch.cuda.set_device(0)
for merge_every in (1, 4):
    r = RankDriver(g, p, 0, world, dims=0, device_index=0, dist=NoDist(), merge_every=merge_every, profile=False)
    r.set_positions(None)
    for k in range(8):
        r.run_iteration(k)
    torch.cuda.synchronize()
    n = 48
    t0 = time.perf_counter()
    r.run_range(list(range(8, 8 + n)))
    t_host = time.perf_counter() - t0                     # launches issued (asynchronous)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    r.profile = True
    r.run_range(list(range(8, 8 + n)))
    tm = r.collect_timing()
    st = r.stats()
    info = r.info
    w = max(tm["windows"], 1)
    print(f"C5 rank 0 of {world}, merge every {merge_every}: quota {info.quota} updates per iteration ({st.n_streams} streams, B {st.bundle}), "
          f"shared slots {info.shared_slots} -> {tm['exchange_bytes_per_window'] / 1e6:.2f} MB per window;\n"
          f"   per window: kernels + pack {tm['compute_ms'] / w * 1e3:.1f} us, apply {tm['exchange_ms'] / w * 1e3:.1f} us (all-reduce: no-op here); "
          f"per iteration: host issue {t_host / n * 1e6:.1f} us, end to end {t_all / n * 1e6:.1f} us "
          f"=> {info.quota * n / t_all / 1e9:.1f} G updates/s on this rank without the collective", flush=True)
    r.close()
