"""DRB1-3123 x120: is the gap between the default sampler and reference streams at short path distances the BUNDLING or the
CONCURRENCY (the team kernels run three streams per 4 nodes, reference streams one per 4)?  Stream-count sweep for both,
three seeds at the default counts.   python scripts/tiled_streams_probe.py"""
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
print("DRB1-3123 x120, -p Y defaults; columns: G upd/s | stress 2M | rel. error at path distance 1, 2-3, 4-7, ... 128-255 | RMSE bp | "
      "d1 without the worst 0.1 % | d1 median", flush=True)


def one(name, flags, T, seed=9399220, extra=""):
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.seed = seed
    ctx.setup_1d(p, hip.make_config(n_streams=T, flags=flags))
    ctx.init_positions()
    ctx.run()
    st = ctx.stats()
    x = ctx.download()
    _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
    lq = Q.layout_quality(g, ctx.sort_order().astype(np.int64))
    sr = Q.short_range_error(g, x, 0, (1,))
    print(f"{name:22s} seed {seed} B {st.bundle:2d} T {st.n_streams:6d} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  "
          f"{O.stress_1d(og, x, 2_000_000):.5f}  " + " ".join(f"{v:.3f}" for v in rms[:8]) + f"  {lq['rmse']:.1f}  {sr['trimmed_rms']:.3f} {sr['median']:.4f}",
          flush=True)


for T in (16384, 32768, 65536, 131072, 262144):
    one("default sampler", 0, T)
for T in (16384, 32768, 65536, 148608, 297216):
    one("reference streams", hip.F_BUNDLE(1), T)
for s in (1, 2):
    one("default sampler", 0, 0, 9399220 + 1000 * s)
    one("reference streams", hip.F_BUNDLE(1), 0, 9399220 + 1000 * s)
one("default, 1 per iter", hip.F_NO_FUSE, 32768)
ctx.close()
