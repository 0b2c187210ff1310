"""windows(200000,16,125000): inversions against the exact stream count around the failing 249 856 (one launch per iteration and
fused), to tell a resonance (a bug-like sharp peak, e.g. at exactly 512 updates per wave and iteration) from a broad occupancy effect.
    python scripts/chain_cap_probe3.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, quality as Q   # noqa: E402

g = G.synth_windows(200_000, 16, 125_000, 7)
ctx = hip.Context(g)
q = P.YgsParams.from_graph(g, 0, 1).path_sgd.min_term_updates
for waves in (2816, 3072, 3328, 3584, 3712, 3840, 3904, 3968, 4032, 4096, 4160, 4352):
    T = waves * 64
    row = []
    for name, flags in (("per-iteration", hip.F_NO_FUSE), ("fused", 0)):
        inv = []
        for s in range(2):
            p = P.YgsParams.from_graph(g, 0, 1).path_sgd
            p.seed = 9399220 + 1000 * s
            ctx.setup_1d(p, hip.make_config(n_streams=T, flags=flags))
            ctx.init_positions()
            ctx.run()
            st = ctx.stats()
            x = ctx.download()
            inv.append((Q.inversions_vs_chain(g.node_ids[ctx.sort_order().astype(np.int64)].astype(np.int64)), float(np.abs(x).max())))
        row.append(f"{name} (launches {st.launches}): " + ", ".join(f"{i} (max |x| {m:.3g})" for i, m in inv))
    print(f"{waves} waves = {T} streams = {T / g.n_nodes:.2f} per node, {q / waves:.1f} updates per wave and iteration: " + " | ".join(row), flush=True)
ctx.close()
