import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip, sgd as S
from oracle import oracle as O
g = G.synth_bubbles(20000, 16, 5)
p = P.LayoutSGDParams.from_graph(g, 2, 1)
og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
c0 = S.default_layout_init(g, 2, p.seed)
for T in (0, 16384, 131072):
    for B in (1, 16, 64):
        res = []
        for k in range(4):
            p.seed = 9399220 + 17 * k
            rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(n_streams=T, flags=hip.F_BUNDLE(B)))
            res.append(O.layout_stress(og, 2, c, 100000))
        print(f"T={st.n_streams} B={B}: " + " ".join(f"{v:.4f}" for v in res), flush=True)
