"""Tapered long runs (sgd_device.h run_taper) on DRB1-3123 x120 and on the 525k-node bubble graph: relative error per octave of
path distance, stress, RMSE, rate, for taper 0 / 8 / 16 / 32 steps against reference streams.   python scripts/taper_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph, oracle_params   # noqa: E402
from gfasort_amd import hip, quality as Q   # noqa: E402


def study(title, g, seeds):
    og = oracle_graph(g)
    ctx = hip.Context(g)
    print(title + "; columns: G upd/s | stress 2M | rel. error at path distance 1, 2-3, 4-7, ... 512-1023 | RMSE bp | d1 without the worst 0.1 %", flush=True)

    def one(name, flags, seed):
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.seed = seed
        ctx.setup_1d(p, hip.make_config(flags=flags))
        ctx.init_positions()
        ctx.run()
        st = ctx.stats()
        x = ctx.download()
        _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
        lq = Q.layout_quality(g, ctx.sort_order().astype(np.int64))
        sr = Q.short_range_error(g, x, 0, (1,))
        print(f"{name:22s} seed {seed} B {st.bundle:2d} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  "
              f"{O.stress_1d(og, x, 2_000_000):.5f}  " + " ".join(f"{v:.4f}" for v in rms[:10]) + f"  {lq['rmse']:.2f}  {sr['trimmed_rms']:.4f}", flush=True)

    for s in seeds:
        os.environ.pop("GFS_DBG_TAPER", None)
        one("reference streams", hip.F_BUNDLE(1), s)
        for t in (0, 8, 16, 32):
            os.environ["GFS_DBG_TAPER"] = str(t)
            one(f"default, taper {t}", 0, s)
    os.environ.pop("GFS_DBG_TAPER", None)
    ctx.close()


study("DRB1-3123 x120, -p Y defaults", G.tile_series(load("DRB1-3123.gfa"), 120), (9399220, 9400220, 9401220))
study("synth_bubbles(400000,24,6), -p Y defaults", G.synth_bubbles(400_000, 24, 6), (9399220,))
