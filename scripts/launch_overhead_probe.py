"""C3 (-p Y), one launch per iteration with fixed quotas: kernel time against the updates per launch — the fixed cost of a launch
(ramp + tail) and the marginal rate.   python scripts/launch_overhead_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip   # noqa: E402

g = G.synth_windows(1_000_000, 64, 156_250, 2)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
p.iter_max = 200
ctx = hip.Context(g)
rows = []
for q in (2_500_000, 5_000_000, 10_000_000, 20_000_000, 40_000_000, 80_000_000):
    for T in (262144, 131072):
        ctx.setup_1d(p, hip.make_config(term_updates_per_iteration=q, n_streams=T))
        ctx.init_positions()
        ctx.run_iteration(0)
        ctx.synchronize()
        s0 = ctx.stats()
        for k in range(1, 21):
            ctx.run_iteration(k)
        ctx.synchronize()
        s1 = ctx.stats()
        us = (s1.kernel_ms - s0.kernel_ms) / 20 * 1e3
        rows.append((T, q, us))
        print(f"streams {T:6d}  {q:9d} updates per launch: {us:8.1f} us = {q / us / 1e3:6.1f} G updates/s", flush=True)
ctx.close()
for T in (262144, 131072):
    qs = np.array([r[1] for r in rows if r[0] == T], dtype=float)
    us = np.array([r[2] for r in rows if r[0] == T])
    b, a = np.polyfit(qs, us, 1)
    print(f"streams {T}: time = {a:.1f} us + updates / {1.0 / b / 1e3:.1f} G per s", flush=True)
