import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip, sgd as S
from oracle import oracle as O
g = G.synth_windows(1_000_000, 64, 156_250, 2)
og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
for dims in (2, 3):
    p = P.LayoutSGDParams.from_graph(g, dims, 1)
    c0 = S.default_layout_init(g, dims, p.seed)
    for B in (1, 16, 64):
        for T in (0, 262144):
            rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(n_streams=T, flags=hip.F_BUNDLE(B)))
            print(f"C4 D={dims} B={B} T={st.n_streams}: {st.term_updates/(st.kernel_ms*1e-3)/1e9:.2f} G upd/s kernel, call {st.total_ms:.0f} ms, stress {O.layout_stress(og, dims, c, 100000):.4e}", flush=True)
print("--- end-distance check")
p = P.LayoutSGDParams.from_graph(g, 2, 1)
c0 = S.default_layout_init(g, 2, p.seed)
for B in (1, 8, 64):
    rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=hip.F_BUNDLE(B)))
    cc = c.reshape(-1, 2, 2)
    d = np.sqrt(((cc[:, 0, :] - cc[:, 1, :]) ** 2).sum(axis=1))
    e = np.abs(d - g.node_len)
    print(f"B={B}: |end distance - node_len| median {np.median(e):.4f} mean {e.mean():.4f} p99 {np.percentile(e,99):.4f} max {e.max():.3f}", flush=True)
