"""DRB1-3123 x120, --iter-max 100: is it the SHORT-jump terms' bundling (64 consecutive steps per trip) that makes the team kernel
fall behind in the cooling half?  (code: git history, commit "Experiments on the cooling-phase lag") GFS_DBG2=4 scattered the terms of short-jump leaders independently over the path (same number
of terms, reference-like); long jumps stay bundled.   python scripts/tiled_short_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph   # noqa: E402
from gfasort_amd import hip, quality as Q   # noqa: E402

for title, g in (("DRB1-3123 x120", G.tile_series(load("DRB1-3123.gfa"), 120)), ("bubbles 525k", G.synth_bubbles(400_000, 24, 6))):
    og = oracle_graph(g)
    ctx = hip.Context(g)
    print(title + ", -p Y --iter-max 100; columns: G upd/s | stress 2M | rel. error at path distance 1, 2-3, ... 512-1023 | RMSE bp", flush=True)
    for name, flags, dbg2 in (("reference streams", hip.F_BUNDLE(1), 0), ("team kernel", 0, 0), ("team, sweeps at half step", 0, 16), ("team, sweeps at half step in cooling", 0, 32)):
        os.environ["GFS_DBG2"] = str(dbg2)
        for seed in (9399220, 9400220):
            p = P.YgsParams.from_graph(g, 0, 1).path_sgd
            p.seed = seed
            ctx.setup_1d(p, hip.make_config(flags=flags))
            ctx.init_positions()
            ctx.run()
            st = ctx.stats()
            x = ctx.download()
            _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
            lq = Q.layout_quality(g, hip.sort_order(x).astype(np.int64))
            print(f"{name:36s} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  {O.stress_1d(og, x, 2_000_000):.5f}  "
                  + " ".join(f"{v:.3f}" for v in rms[:10]) + f"  {lq['rmse']:.1f}", flush=True)
    os.environ.pop("GFS_DBG2", None)
    ctx.close()
