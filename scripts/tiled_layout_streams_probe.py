"""DRB1-3123 x120, `-p L --dimensions 2 --layout-iter 90`: the default layout kernel per stream count, against reference streams
(relative error per octave of path distance as a ratio to reference streams).   python scripts/tiled_layout_streams_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph   # noqa: E402
from gfasort_amd import hip, quality as Q, sgd as S   # noqa: E402

g = G.tile_series(load("DRB1-3123.gfa"), 120)
og = oracle_graph(g)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 90
ref = None
print(f"DRB1-3123 x120 ({g.n_nodes} nodes), -p L --dimensions 2 --layout-iter {ITERS}; columns: streams | G upd/s | layout stress 2M | "
      "rel. error per octave (1, 2-3, ... 512-1023) as a ratio to reference streams", flush=True)
for name, flags, T in (("reference streams", hip.F_BUNDLE(1), 0), ("default", 0, 65536), ("default", 0, 131072), ("default", 0, 131072),
                       ("default", 0, 196608), ("default", 0, 196608), ("default", 0, 262144), ("one partner", hip.F_ONE_PARTNER, 196608),
                       ("K = 64", hip.F_CHAIN(64), 196608)):
    p = P.LayoutSGDParams.from_graph(g, 2, 1)
    p.iter_max = ITERS
    c0 = S.default_layout_init(g, 2, p.seed)
    rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=flags, n_streams=T))
    _, rms, _ = Q.stress_by_scale(g, c, 2, 1_000_000)
    if ref is None:
        ref = rms
    print(f"{name:18s} {st.n_streams:7d} K {st.run_trips:2d} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  "
          f"{O.layout_stress(og, 2, c, 2_000_000):.5f}  " + " ".join(f"{v:.3f}" for v in (rms / ref)[:10]), flush=True)
