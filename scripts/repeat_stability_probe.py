"""Reference streams on the tandem-repeat graphs of tests/test_gpu_parity.py (crowded nodes, stable only just at the
streams-per-node bound): one fused pooled launch against one launch per iteration, 5 seeds each — final sampled stress.
    python scripts/repeat_stability_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, oracle_graph, oracle_params   # noqa: E402
from gfasort_amd import hip   # noqa: E402

for period, copies, every in ((1, 40, 500), (5, 20, 300), (1, 200, 2000)):
    g = G.synth_repeats(60_000, 16, period, copies, every, 3)
    og = oracle_graph(g)
    p0 = P.YgsParams.from_graph(g, 0, 1).path_sgd
    x_ref = O.init_positions(og)
    s0 = O.stress_1d(og, x_ref, 100000)
    O.sgd_1d(og, oracle_params(p0), x_ref, n_streams=8)
    print(f"repeats(period {period}, copies {copies}, every {every}): start {s0:.4f}, oracle {O.stress_1d(og, x_ref, 100000):.4f}", flush=True)
    for name, flags in (("reference streams fused", hip.F_BUNDLE(1)), ("reference streams per iteration", hip.F_BUNDLE(1) | hip.F_NO_FUSE), ("default", 0)):
        vals = []
        for k in range(5):
            p = P.YgsParams.from_graph(g, 0, 1).path_sgd
            p.seed = 9399220 + 1000 * k
            rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(flags=flags))
            vals.append(O.stress_1d(og, x, 100000) if np.isfinite(x).all() else float("nan"))
        print(f"    {name:32s} B {st.bundle:2d} T {st.n_streams:6d}: " + " ".join(f"{v:.4g}" for v in vals), flush=True)
