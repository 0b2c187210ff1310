import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from ablate2 import run

def permuted(g, perm):
    """relabel dense indices: new index perm[k] for old index k"""
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    return G.FlatGraph(node_len=g.node_len[inv], step_node=perm[g.step_node].astype(np.uint32), step_is_rev=g.step_is_rev,
                       path_first_step=g.path_first_step, node_ids=g.node_ids[inv], path_names=g.path_names, step_node_id=g.step_node_id)

g0 = G.synth_windows(1_000_000, 64, 156_250, 2, shuffle=False)
rng = np.random.default_rng(0)
variants = [("sorted", g0), ("block-shuffled(64)", G.synth_windows(1_000_000, 64, 156_250, 2, shuffle=True)),
            ("random order", permuted(g0, rng.permutation(g0.n_nodes)))]
for name, g in variants:
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = 200
    for B in (1, 64):
        r, _ = run(g, p, hip.F_BUNDLE(B), k0=3)
        print(f"C3 node order {name:20s} B={B:2d}: {r:7.3f} G/s", flush=True)
