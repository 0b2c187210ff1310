"""How much would the team kernel gain on bubble graphs from a node layout that keeps both alleles of a bubble next to each
other?  Upper bound: lay the nodes out in the order of a finished sort (gfs_ctx_create_with_layout), run again."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip

for name, g in (("bubbles 525k", G.synth_bubbles(400_000, 24, 6)), ("bubbles 2M", G.synth_bubbles(1_500_000, 32, 7))):
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    ctx = hip.Context(g)
    ctx.setup_1d(p, hip.make_config())
    ctx.init_positions(); ctx.run()
    st = ctx.stats(); x = ctx.download(); ctx.close()
    print(name, "first-visit layout:", f"{st.term_updates / (st.kernel_ms * 1e-3) / 1e9:.2f} G updates/s", flush=True)
    for iters in (5, 10, 20, 100):
        p2 = P.YgsParams.from_graph(g, 0, 1).path_sgd
        ctx = hip.Context(g)
        ctx.setup_1d(p2, hip.make_config())
        ctx.init_positions(); ctx.run_range(list(range(iters))); xs = ctx.download(); ctx.close()
        perm = np.empty(g.n_nodes, dtype=np.uint32)
        perm[np.argsort(xs, kind="stable")] = np.arange(g.n_nodes, dtype=np.uint32)       # dense index -> slot = rank by position
        ctx = hip.Context(g, node_perm=perm)
        ctx.setup_1d(p, hip.make_config())
        ctx.init_positions(); ctx.run()
        st = ctx.stats(); ctx.close()
        print(name, f"layout = order after {iters:3d} iterations:", f"{st.term_updates / (st.kernel_ms * 1e-3) / 1e9:.2f} G updates/s", flush=True)
