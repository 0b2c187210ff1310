"""C3-sized graph cut into many short paths: path/zeta tables in LDS (<= 3072 paths) or in global memory."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip

def main():
    for paths in (64, 1024, 3000, 4096, 16384):
        w = 10_000_000 // paths
        g = G.synth_windows(1_000_000, paths, w, 2)
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = 100
        rc, x, st = hip.path_linear_sgd_raw(g, p)
        ids = g.node_ids[np.argsort(x, kind="stable")].astype(np.int64)
        ok = bool(np.all(np.diff(ids) == 1) or np.all(np.diff(ids) == -1))
        print(f"paths {paths:6d} steps/path {w:7d}: bundle {st.bundle} streams {st.n_streams} launches {st.launches} "
              f"{st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.2f} G upd/s  order_ok={ok}", flush=True)

if __name__ == "__main__":
    main()
