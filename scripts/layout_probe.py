"""What the internal first-visit node layout buys: C3 with its nodes in random input order, positions stored in
first-visit path order (default) or in input order (explicit identity layout)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from order_probe_util import permuted

def rate(g, p, node_perm):
    ctx = hip.Context(g, node_perm=node_perm)
    ctx.setup_1d(p, hip.make_config())
    ctx.upload(hip.init_positions(g))
    ctx.run()
    st = ctx.stats()
    x = ctx.download()
    ctx.close()
    ids = g.node_ids[np.argsort(x, kind="stable")].astype(np.int64)
    ok = bool(np.all(np.diff(ids) == 1) or np.all(np.diff(ids) == -1))
    return st.term_updates / (st.kernel_ms * 1e-3) / 1e9, ok

def main():
    g0 = G.synth_windows(1_000_000, 64, 156_250, 2, shuffle=False)
    g = permuted(g0, np.random.default_rng(0).permutation(g0.n_nodes))
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = 200
    for name, perm in (("first-visit path order (default)", None), ("input order (identity layout)", np.arange(g.n_nodes, dtype=np.uint32))):
        r, ok = rate(g, p, perm)
        print(f"C3, nodes in random input order, positions in {name:34s}: {r:6.2f} G updates/s  order_ok={ok}", flush=True)

if __name__ == "__main__":
    main()
