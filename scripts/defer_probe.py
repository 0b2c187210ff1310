"""Streams vs deferral on graphs of 400k-1M nodes: the team kernel defers its atomics only while 4*streams <= nodes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip

def main():
    for n, paths in ((400_000, 64), (600_000, 64), (800_000, 64), (1_000_000, 64)):
        g = G.synth_windows(n, paths, n * 10 // paths, 2)
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = 200
        for T in (0, (n // 4) // 64 * 64, 131072, 196608, 249856):
            if T > n // 2:
                continue
            rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(n_streams=T))
            ids = g.node_ids[np.argsort(x, kind="stable")].astype(np.int64)
            ok = bool(np.all(np.diff(ids) == 1) or np.all(np.diff(ids) == -1))
            print(f"nodes {n:8d} streams {'auto' if T == 0 else T:>7} -> {st.n_streams:6d}  defer {'on ' if 4 * st.n_streams <= n else 'off'}: "
                  f"{st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.2f} G upd/s  order_ok={ok}", flush=True)

if __name__ == "__main__":
    main()
