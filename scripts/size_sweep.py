"""Team kernel (auto policy, fused run) across graph sizes: windows graphs with 10 steps per node."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip

def main():
    print("# nodes  paths  steps  iters  streams bundle  G upd/s  frac of 8 TB/s (64 B/update)  ctx_create s  order_ok", flush=True)
    for n, paths in ((20_000, 16), (100_000, 32), (300_000, 64), (1_000_000, 64), (3_000_000, 128), (10_000_000, 256), (30_000_000, 512)):
        w = n * 10 // paths
        g = G.synth_windows(n, paths, min(w, n), 2)
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = 200 if n <= 1_000_000 else (60 if n <= 10_000_000 else 20)
        t0 = time.time()
        ctx = hip.Context(g)
        t_ctx = time.time() - t0
        ctx.setup_1d(p, hip.make_config())
        ctx.upload(hip.init_positions(g))
        ctx.run()
        st = ctx.stats()
        x = ctx.download()
        ctx.close()
        ids = g.node_ids[np.argsort(x, kind="stable")].astype(np.int64)
        ok = bool(np.all(np.diff(ids) == 1) or np.all(np.diff(ids) == -1))
        rate = st.term_updates / (st.kernel_ms * 1e-3)
        print(f"{n:9d} {paths:4d} {g.n_steps:10d} {p.iter_max:4d} {st.n_streams:7d} {st.bundle:3d}  {rate / 1e9:7.2f}  {rate * 64 / 8e12:6.3f}  "
              f"{t_ctx:6.2f}  {ok if p.iter_max >= 200 else 'n/a (short run)'}", flush=True)

if __name__ == "__main__":
    main()
