"""EXPERIMENT (needs the kernel code of the commit "Experiment: clipping the differential part ..."; the current library
ignores GFS_DBG2): DRB1-3123 x120 (-p Y --iter-max 100) and the 525k-node bubble graph with the differential part of a twin trip's
errors clipped (GFS_DBG2 = 512 | kappa << 12).   python scripts/tiled_clip_probe.py"""
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
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    print(f"{title}, -p Y --iter-max 100; columns: G upd/s | stress 2M | rel. error at path distance 1, 2-3, ... 512-1023 | d1 trimmed | RMSE bp", flush=True)
    VAR = (("reference streams", hip.F_BUNDLE(1), 0), ("default", 0, 0), ("default", 0, 0), ("clip kappa 1", 0, 512 | (1 << 12)),
           ("clip kappa 2", 0, 512 | (2 << 12)), ("clip kappa 4", 0, 512 | (4 << 12)), ("clip kappa 4", 0, 512 | (4 << 12)), ("clip kappa 8", 0, 512 | (8 << 12)))
    if "--permute" in sys.argv:
        VAR = (("reference streams", hip.F_BUNDLE(1), 0), ("default", 0, 0), ("default", 0, 0), ("partners permuted", 0, 1024), ("partners permuted", 0, 1024),
               ("partners permuted", 0, 1024))
    for name, flags, dbg2 in VAR:
        os.environ["GFS_DBG2"] = str(dbg2)
        ctx = hip.Context(g)
        ctx.setup_1d(p, hip.make_config(flags=flags)); ctx.init_positions(); ctx.run()
        x, st = ctx.download(), ctx.stats(); ctx.close()
        _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
        sr = Q.short_range_error(g, x, 0, (1,))
        q = Q.layout_quality(g, hip.sort_order(x).astype(np.int64))
        print(f"{name:20s} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  {O.stress_1d(og, x, 2_000_000):.5f}  " +
              " ".join(f"{v:.3f}" for v in rms[:10]) + f"  {sr['trimmed_rms']:.2f}  {q['rmse']:.1f}", flush=True)
    os.environ.pop("GFS_DBG2", None)
