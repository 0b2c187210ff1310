import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import *
from gfasort_amd import hip
import numpy as np
rng = np.random.default_rng(5)
n = 6000
lens = rng.integers(1, 9, n).astype(np.uint32)
long_path = np.arange(n, dtype=np.uint32)
shorts = [np.arange(s, s + 12, dtype=np.uint32) for s in rng.integers(0, n - 12, 4000)]
steps = np.concatenate([long_path] + shorts)
firsts = np.concatenate([[0], np.cumsum([len(long_path)] + [12] * len(shorts))]).astype(np.uint64)
rev = (rng.random(steps.shape[0]) < 0.3).astype(np.uint8)
g = G.FlatGraph(node_len=lens, step_node=steps, step_is_rev=rev, path_first_step=firsts, node_ids=np.arange(1, n + 1, dtype=np.uint64), path_names=[f"p{k}" for k in range(len(shorts) + 1)])
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
og = oracle_graph(g)
for iters in (30, 100):
    p.iter_max = iters
    for b in (1, 0, 8, 64):
        for T in (0, 6784, 1024, 64):
            x0 = np.asarray(rng.permutation(n), dtype=np.float64) * 4.5
            rc, x, st = hip.path_linear_sgd_raw(g, p, x=x0.copy(), cfg=hip.make_config(n_streams=T, flags=hip.F_BUNDLE(b)))
            print(f"iters {iters} bundle req {b} used {st.bundle} T {st.n_streams}: stress {O.stress_1d(og, x0, 50000):.1f} -> {O.stress_1d(og, x, 50000):.4g} updates {st.term_updates} att {st.attempts}", flush=True)
