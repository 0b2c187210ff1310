"""`-p L --dimensions 2` at the CLI's default schedule (--layout-iter 30): two partners per leader (the default) against one, against
reference streams — DRB1-3123 x120 and the 525k-node bubble graph, two seeds.   python scripts/layout_partner_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph   # noqa: E402
from gfasort_amd import hip, quality as Q, sgd as S   # noqa: E402

for title, g in (("DRB1-3123 x120", G.tile_series(load("DRB1-3123.gfa"), 120)), ("bubbles 525k", G.synth_bubbles(400_000, 24, 6))):
    og = oracle_graph(g)
    print(title + ", -p L --dimensions 2 --layout-iter 30; columns: G upd/s | layout stress 2M | rel. error at path distance 1, 2-3, ... 512-1023 | "
          "median |end-to-end - length|", flush=True)
    ref = None
    for name, flags in (("reference streams", hip.F_BUNDLE(1)), ("two partners (default)", 0), ("one partner", hip.F_ONE_PARTNER)):
        profs = []
        for seed in (9399220, 9400220):
            p = P.LayoutSGDParams.from_graph(g, 2, 1)
            p.seed = seed
            c0 = S.default_layout_init(g, 2, p.seed)
            rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=flags))
            _, rms, _ = Q.stress_by_scale(g, c, 2, 1_000_000)
            profs.append(rms[:10])
            cc = c.reshape(-1, 2, 2)
            e2e = np.median(np.abs(np.sqrt(((cc[:, 0] - cc[:, 1]) ** 2).sum(axis=1)) - g.node_len))
            print(f"{name:24s} seed {seed} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  {O.layout_stress(og, 2, c, 2_000_000):.5f}  "
                  + " ".join(f"{v:.4f}" for v in rms[:10]) + f"  {e2e:.3f}", flush=True)
        m = np.mean(profs, axis=0)
        if ref is None:
            ref = m
        else:
            print(f"{'':24s} ratio to reference streams (mean of the seeds): " + " ".join(f"{v:.3f}" for v in m / ref), flush=True)
