"""Exact chain order (P1) of window graphs against the team kernel's streams per node: 3 seeds per cell, adjacent inversions of
the final sort (0 = exact).   python scripts/chain_cap_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, quality as Q   # noqa: E402

print("graph | streams per node -> streams: inversions per seed", flush=True)
for n, paths, w in ((66_000, 16, 41_000), (131_000, 16, 33_000), (200_000, 16, 125_000), (300_000, 32, 60_000)):
    g = G.synth_windows(n, paths, w, 7)
    ctx = hip.Context(g)
    for per_node in (0.75, 1.0, 1.25):
        T = int(g.n_nodes * per_node) // 256 * 256
        inv = []
        for s in range(3):
            p = P.YgsParams.from_graph(g, 0, 1).path_sgd
            p.seed = 9399220 + 1000 * s
            ctx.setup_1d(p, hip.make_config(n_streams=T))
            ctx.init_positions()
            ctx.run()
            st = ctx.stats()
            inv.append(Q.inversions_vs_chain(g.node_ids[ctx.sort_order().astype(np.int64)].astype(np.int64)))
        print(f"windows({n},{paths},{w}) | {per_node:4.2f} -> {st.n_streams:6d} (B {st.bundle} K {st.run_trips}): {inv}", flush=True)
    ctx.close()
