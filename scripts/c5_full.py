"""BASELINE configs[4] on one GPU: the whole default -p Y run of the 10M-node / 1e8-step graph, with the order check."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
t0 = time.time()
g = G.synth_windows(10_000_000, 1024, 97_656, 3)
print("graph", time.time() - t0, g.n_steps, flush=True)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
print("iter_max", p.iter_max, "M", p.min_term_updates, flush=True)
t0 = time.time()
rc, x, st = hip.path_linear_sgd_raw(g, p)
print("call", time.time() - t0, st.term_updates, st.kernel_ms, st.launches, st.bundle, flush=True)
ids = g.node_ids[np.argsort(x, kind="stable")].astype(np.int64)
d = np.diff(ids)
print("order ok", bool(np.all(d == 1) or np.all(d == -1)), "inversions", int((d != (1 if ids[0] < ids[-1] else -1)).sum()), flush=True)
