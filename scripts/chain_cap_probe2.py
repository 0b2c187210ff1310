"""The densely covered window graph that lost its exact chain order above three streams per 4 nodes: is it the streams per node, or
the pooled launch's spread over iterations (an iteration of this graph is 976 chunks of 2048 updates for up to 3904 waves)?
One launch per iteration (no spread) against the fused pooled launch.   python scripts/chain_cap_probe2.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, quality as Q   # noqa: E402

g = G.synth_windows(200_000, 16, 125_000, 7)
ctx = hip.Context(g)
print("windows(200000,16,125000): inversions per seed; quota", P.YgsParams.from_graph(g, 0, 1).path_sgd.min_term_updates, flush=True)
for name, flags in (("fused pooled launch", 0), ("one launch per iteration", hip.F_NO_FUSE)):
    for per_node in (0.75, 1.0, 1.25, 1.5, 2.0):
        T = int(g.n_nodes * per_node) // 256 * 256
        inv = []
        for s in range(3):
            p = P.YgsParams.from_graph(g, 0, 1).path_sgd
            p.seed = 9399220 + 1000 * s
            ctx.setup_1d(p, hip.make_config(n_streams=T, flags=flags))
            ctx.init_positions()
            ctx.run()
            st = ctx.stats()
            inv.append(Q.inversions_vs_chain(g.node_ids[ctx.sort_order().astype(np.int64)].astype(np.int64)))
        print(f"{name:26s} {per_node:4.2f} per node -> {st.n_streams:6d} streams = {st.n_streams // 64} waves: {inv}", flush=True)
ctx.close()
