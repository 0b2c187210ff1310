"""Rate and quality on the pangenome-like bubble graphs (SNP bubbles + insertions), final kernels, auto policy."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from oracle import oracle as O

for name, g in (("bubbles 400k sites x 24 hap", G.synth_bubbles(400000, 24, 6)), ("bubbles 1.5M sites x 32 hap", G.synth_bubbles(1_500_000, 32, 7))):
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
    s0 = O.stress_1d(og, O.init_positions(og), 200000)
    for B in (1, 0):
        rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(flags=hip.F_BUNDLE(B)))
        print(f"{name}: nodes {g.n_nodes} steps {g.n_steps} bundle {st.bundle} streams {st.n_streams}: "
              f"{st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.2f} G upd/s, call {st.total_ms:.0f} ms, stress {s0:.3g} -> {O.stress_1d(og, x, 200000):.4g}", flush=True)
