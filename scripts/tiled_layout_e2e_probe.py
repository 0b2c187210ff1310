"""DRB1-3123 x120, `-p L --dimensions 2 --layout-iter 90`: |distance between a node's two ends - its length| (median, mean) and the
layout stress per launch mode of the default kernel, four seeds each.   python scripts/tiled_layout_e2e_probe.py [iters]"""
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
print(f"DRB1-3123 x120, -p L --dimensions 2 --layout-iter {ITERS}; columns: seed | launches | stress 2M | rel. error at distance 1, 2-3, 4-7 | "
      "end-to-end median, mean", flush=True)
for name, flags in (("reference streams", hip.F_BUNDLE(1)), ("default (fused, pools)", 0), ("one launch per iteration", hip.F_NO_FUSE),
                    ("fused, fixed quotas", hip.F_DBG_FREE_RUNNING)):
    acc = []
    for seed in range(4):
        p = P.LayoutSGDParams.from_graph(g, 2, 1)
        p.iter_max = ITERS
        p.seed = p.seed + 1000 * seed
        c0 = S.default_layout_init(g, 2, p.seed)
        rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=flags))
        _, rms, _ = Q.stress_by_scale(g, c, 2, 1_000_000)
        cc = np.asarray(c).reshape(-1, 2, 2)
        err = np.abs(np.sqrt(((cc[:, 0, :] - cc[:, 1, :]) ** 2).sum(axis=1)) - g.node_len)
        row = [O.layout_stress(og, 2, c, 2_000_000), rms[0], rms[1], rms[2], float(np.median(err)), float(np.mean(err))]
        acc.append(row)
        print(f"{name:26s} seed +{1000 * seed:4d} launches {st.launches:3d}  " + " ".join(f"{v:.4f}" for v in row), flush=True)
    print(f"{name:26s} mean                       " + " ".join(f"{v:.4f}" for v in np.mean(acc, axis=0)), flush=True)
