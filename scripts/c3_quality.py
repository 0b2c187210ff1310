import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from oracle import oracle as O
g = G.synth_windows(1_000_000, 64, 156_250, 2)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
p.iter_max = 200
og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
for B in (1, 16, 64):
    for T in (131072, 524288):
        for seed in (9399220, 1):
            p.seed = seed
            rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(n_streams=T, flags=hip.F_BUNDLE(B)))
            order = hip.sort_order(x).astype(np.int64)
            ids = g.node_ids[order].astype(np.int64)
            fwd = ids if ids[0] < ids[-1] else ids[::-1]
            inv = int((np.diff(fwd) < 0).sum())
            disp = int(np.abs(fwd - np.arange(1, len(fwd) + 1)).max())
            gaps = np.abs(np.diff(x[order]))
            print(f"B={B} T={T} seed={seed}: updates {st.term_updates} stress {O.stress_1d(og, x, 200000):.3e} inversions {inv} max displacement {disp} "
                  f"min gap {gaps.min():.3e} finite {np.isfinite(x).all()}", flush=True)
