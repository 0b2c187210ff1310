"""Default flags on real pangenome structure: tests/data/DRB1-3123.gfa tiled 120x in series (gfasort_amd/graph.py
tile_series; 594 600 nodes, 12 paths of up to 372 000 steps), `-p Y` defaults.  Relative error per octave of path
distance (distance 1 and 2-3 over ALL pairs), sampled stress, for the library's default, reference streams and the
sampler's variants — which ingredient of the default sampler costs what on this graph.
    python scripts/tiled_probe.py [copies = 120] [shuffle_seed = -1] [variant-set = all]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph, oracle_params   # noqa: E402
from gfasort_amd import hip, quality as Q   # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 120
shuffle = int(sys.argv[2]) if len(sys.argv) > 2 else -1
which = sys.argv[3] if len(sys.argv) > 3 else "all"
g = G.tile_series(load("DRB1-3123.gfa"), copies, shuffle_seed=None if shuffle < 0 else shuffle)
og = oracle_graph(g)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
print(f"DRB1-3123 x{copies} (shuffle {shuffle}): {g.n_nodes} nodes, {g.n_steps} steps, {g.n_paths} paths, -p Y --iter-max {p.iter_max}; "
      f"columns: G upd/s | stress 2M | rel. error at path distance 1, 2-3, 4-7, ... 512-1023 | RMSE bp", flush=True)
variants = [("reference streams", hip.F_BUNDLE(1)), ("default", 0),
            ("B=64 K=1", hip.F_CHAIN(1)), ("B=64 K=4", hip.F_CHAIN(4)), ("B=64 K=16", hip.F_CHAIN(16)),
            ("one partner", hip.F_ONE_PARTNER), ("no twin trips", hip.F_DBG_NO_TWIN_TRIP), ("no fused trips", hip.F_DBG_NO_FUSED_TRIP),
            ("no line alignment", hip.F_DBG_NO_ALIGN),
            ("one launch per iteration", hip.F_NO_FUSE), ("B=16", hip.F_BUNDLE(16)), ("B=32", hip.F_BUNDLE(32)),
            ("K=1 one partner no fused", hip.F_CHAIN(1) | hip.F_ONE_PARTNER | hip.F_DBG_NO_FUSED_TRIP)]
if which == "few":
    variants = variants[:2]
ctx = hip.Context(g)
for name, flags in variants:
    ctx.setup_1d(p, hip.make_config(flags=flags))
    ctx.init_positions()
    ctx.run()
    st = ctx.stats()
    x = ctx.download()
    _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
    lq = Q.layout_quality(g, ctx.sort_order().astype(np.int64))
    print(f"{name:28s} B {st.bundle:2d} K {st.run_trips:2d} T {st.n_streams:6d} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  "
          f"{O.stress_1d(og, x, 2_000_000):.5f}  " + " ".join(f"{v:.4f}" for v in rms[:10]) + f"  {lq['rmse']:.2f}", flush=True)
ctx.close()
op = oracle_params(p)
op.nthreads = max(2, min(16, len(os.sched_getaffinity(0))))
x = O.init_positions(og)
rc, cst = O.sgd_1d_threads(og, op, x, flat=1)
_, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
lq = Q.layout_quality(g, hip.sort_order(x).astype(np.int64))
print(f"{'CPU oracle, threads':28s} {'':24s} {O.stress_1d(og, x, 2_000_000):.5f}  " + " ".join(f"{v:.4f}" for v in rms[:10]) + f"  {lq['rmse']:.2f}", flush=True)
