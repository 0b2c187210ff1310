import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from oracle import oracle as O
g = G.synth_bubbles(1_500_000, 32, 7)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
for label, flags, T in (("B=1", hip.F_BUNDLE(1), 0), ("B=8", hip.F_BUNDLE(8), 0), ("B=16", hip.F_BUNDLE(16), 0), ("B=32", hip.F_BUNDLE(32), 0), ("B=64", hip.F_BUNDLE(64), 0),
                        ("B=64 no align", hip.F_BUNDLE(64) | 0x1000, 0), ("B=64 T=131072", hip.F_BUNDLE(64), 131072), ("B=64 no defer", hip.F_BUNDLE(64) | 0x400, 0)):
    res = []
    for sd in range(2):
        p.seed = 9399220 + 1000 * sd
        rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(n_streams=T, flags=flags))
        res.append(O.stress_1d(og, x, 200000))
    print(f"bubbles 1.5M {label:16s} streams {st.n_streams}: {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.2f} G upd/s  stress " + " ".join(f"{v:.4g}" for v in res), flush=True)
