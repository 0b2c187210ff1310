"""DRB1-3123 x120, `-p L --dimensions 2`: the default layout kernel against reference streams at the CLI's schedule
(--layout-iter 30) and at three times that.   python scripts/tiled_layout_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph, oracle_params   # noqa: E402
from gfasort_amd import hip, quality as Q, sgd as S   # noqa: E402

g = G.tile_series(load("DRB1-3123.gfa"), 120)
og = oracle_graph(g)
print("DRB1-3123 x120, -p L --dimensions 2; columns: G upd/s | layout stress 2M | rel. error at path distance 1, 2-3, ... 512-1023 | "
      "median |end-to-end - length|", flush=True)
for iters in (30, 90):
    for name, flags in (("reference streams", hip.F_BUNDLE(1)), ("default", 0), ("default K = 64", hip.F_CHAIN(64)), ("one partner", hip.F_ONE_PARTNER)):
        p = P.LayoutSGDParams.from_graph(g, 2, 1)
        p.iter_max = iters
        c0 = S.default_layout_init(g, 2, p.seed)
        rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=flags))
        _, rms, _ = Q.stress_by_scale(g, c, 2, 1_000_000)
        cc = c.reshape(-1, 2, 2)
        e2e = np.median(np.abs(np.sqrt(((cc[:, 0] - cc[:, 1]) ** 2).sum(axis=1)) - g.node_len))
        print(f"{name:20s} iters {iters:3d} B {st.bundle:2d} K {st.run_trips:2d} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  "
              f"{O.layout_stress(og, 2, c, 2_000_000):.5f}  " + " ".join(f"{v:.4f}" for v in rms[:10]) + f"  {e2e:.3f}", flush=True)
