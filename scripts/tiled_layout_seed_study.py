"""DRB1-3123 x120, `-p L --dimensions 2 --layout-iter 90`: how much one run of either sampler differs from the next.  Four seeds
each of reference streams and of the default kernel; per run the layout stress and the relative error per octave of path distance.
    python scripts/tiled_layout_seed_study.py [iters]"""
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
print(f"DRB1-3123 x120 ({g.n_nodes} nodes), -p L --dimensions 2 --layout-iter {ITERS}; columns: seed | streams | layout stress 2M | "
      "rel. error per octave (1, 2-3, ... 512-1023)", flush=True)
prof = {}
for name, flags in (("reference streams", hip.F_BUNDLE(1)), ("default", 0)):
    for seed in (0, 1, 2, 3):
        p = P.LayoutSGDParams.from_graph(g, 2, 1)
        p.iter_max = ITERS
        p.seed = p.seed + 1000 * seed
        c0 = S.default_layout_init(g, 2, p.seed)
        rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=flags))
        _, rms, _ = Q.stress_by_scale(g, c, 2, 1_000_000)
        prof.setdefault(name, []).append(rms[:10])
        print(f"{name:18s} seed +{1000 * seed:4d} {st.n_streams:7d}  {O.layout_stress(og, 2, c, 2_000_000):.5f}  " +
              " ".join(f"{v:.3f}" for v in rms[:10]), flush=True)
for name, rows in prof.items():
    a = np.array(rows)
    print(f"{name:18s} mean over seeds        " + " ".join(f"{v:.3f}" for v in a.mean(axis=0)), flush=True)
    print(f"{name:18s} sd / mean              " + " ".join(f"{v:.3f}" for v in a.std(axis=0) / a.mean(axis=0)), flush=True)
r = np.array(prof["default"]).mean(axis=0) / np.array(prof["reference streams"]).mean(axis=0)
print("default / reference streams, means " + " ".join(f"{v:.3f}" for v in r), flush=True)
