"""DRB1-3123 x120, --iter-max 100: WHERE in the schedule does the bundled sampler fall behind reference streams?  The schedule is
split at iteration k0: one sampler before it, the other after (positions carried over), every combination and several k0.
    python scripts/tiled_phase_probe.py"""
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
REF, TEAM = hip.F_BUNDLE(1), 0
print("DRB1-3123 x120, -p Y --iter-max 100; columns: kernel ms | stress 2M | rel. error at path distance 1, 2-3, ... 512-1023 | RMSE bp", flush=True)


def run(name, phases):
    x = None
    ms = 0.0
    for flags, ks in phases:
        if not len(ks):
            continue
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
    print(f"{name:44s} {ms:7.2f}  {O.stress_1d(og, x, 2_000_000):.5f}  " + " ".join(f"{v:.3f}" for v in rms[:10]) + f"  {lq['rmse']:.1f}", flush=True)


n = int(p.iter_max) + 1
run("reference streams throughout", [(REF, range(n))])
run("team kernel throughout", [(TEAM, range(n))])
for k0 in (5, 15, 30, 51):
    run(f"reference streams for k < {k0}, then team", [(REF, range(k0)), (TEAM, range(k0, n))])
for k0 in (51, 70, 85, 95):
    run(f"team for k < {k0}, then reference streams", [(TEAM, range(k0)), (REF, range(k0, n))])
run("team, reference streams for 40 <= k < 70", [(TEAM, range(40)), (REF, range(40, 70)), (TEAM, range(70, n))])
ctx.close()
