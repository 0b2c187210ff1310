"""DRB1-3123 x120: longer runs (K = 128) and longer schedules (--iter-max 300) — does the gap between the default sampler and
reference streams close when runs get longer / when both converge further?   python scripts/tiled_probe2.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph, oracle_params   # noqa: E402
from gfasort_amd import hip, quality as Q   # noqa: E402

g = G.tile_series(load("DRB1-3123.gfa"), 120)
og = oracle_graph(g)
ctx = hip.Context(g)
print("DRB1-3123 x120; columns: G upd/s | stress 2M | rel. error at path distance 1, 2-3, 4-7, ... 512-1023 | RMSE bp | d1 without the worst 0.1 % | d1 median", flush=True)


def one(name, flags, iter_max=100, seed=9399220):
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.seed = seed
    p.iter_max = iter_max
    ctx.setup_1d(p, hip.make_config(flags=flags))
    ctx.init_positions()
    ctx.run()
    st = ctx.stats()
    x = ctx.download()
    _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
    lq = Q.layout_quality(g, ctx.sort_order().astype(np.int64))
    sr = Q.short_range_error(g, x, 0, (1,))
    print(f"{name:26s} iter_max {iter_max:4d} K {st.run_trips:3d} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  "
          f"{O.stress_1d(og, x, 2_000_000):.5f}  " + " ".join(f"{v:.4f}" for v in rms[:10]) + f"  {lq['rmse']:.2f}  {sr['trimmed_rms']:.3f} {sr['median']:.4f}", flush=True)


for s in (9399220, 9400220):
    one("reference streams", hip.F_BUNDLE(1), 100, s)
    one("default (K = 64)", 0, 100, s)
    one("K = 128", hip.F_CHAIN(128), 100, s)
    one("K = 32", hip.F_CHAIN(32), 100, s)
for it in (300, 1000):
    one("reference streams", hip.F_BUNDLE(1), it)
    one("default (K = 64)", 0, it)
ctx.close()
