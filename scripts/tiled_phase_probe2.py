"""DRB1-3123 x120, --iter-max 100: the team kernel with LONG runs (K = 64) before the cooling half and SHORTER runs in it —
do the long-jump runs of the cooling half cause the lag at 32+ steps?   python scripts/tiled_phase_probe2.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph   # noqa: E402
from gfasort_amd import hip, quality as Q   # noqa: E402

g = G.tile_series(load("DRB1-3123.gfa"), 120)
og = oracle_graph(g)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
ctx = hip.Context(g)
print("DRB1-3123 x120, -p Y --iter-max 100; columns: kernel ms | stress 2M | rel. error at path distance 1, 2-3, ... 512-1023 | RMSE bp", flush=True)


def run(name, phases):
    x = None
    ms = 0.0
    for flags, ks in phases:
        ctx.setup_1d(p, hip.make_config(flags=flags))
        if x is None:
            ctx.init_positions()
        else:
            ctx.upload(x)
        s0 = ctx.stats().kernel_ms
        ctx.run_range(list(ks))
        ctx.synchronize()
        ms += ctx.stats().kernel_ms - s0
        x = ctx.download()
    _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
    lq = Q.layout_quality(g, hip.sort_order(x).astype(np.int64))
    print(f"{name:52s} {ms:7.2f}  {O.stress_1d(og, x, 2_000_000):.5f}  " + " ".join(f"{v:.3f}" for v in rms[:10]) + f"  {lq['rmse']:.1f}", flush=True)


n = int(p.iter_max) + 1
run("reference streams throughout", [(hip.F_BUNDLE(1), range(n))])
run("team kernel throughout (K = 64)", [(0, range(n))])
for k in (1, 4, 16):
    run(f"team K = 64 for k < 51, K = {k} in the cooling half", [(0, range(51)), (hip.F_CHAIN(k), range(51, n))])
run("team K = 64 for k < 51, one partner K = 1 after", [(0, range(51)), (hip.F_CHAIN(1) | hip.F_ONE_PARTNER, range(51, n))])
run("team K = 64 for k < 51, B = 16 after", [(0, range(51)), (hip.F_BUNDLE(16), range(51, n))])
ctx.close()
