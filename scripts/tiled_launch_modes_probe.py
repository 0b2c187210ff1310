"""DRB1-3123 x120, -p Y --iter-max 100: the lag of the run sampler per launch mode (default) or per sampler variant (--samplers).
    python scripts/tiled_launch_modes_probe.py [--samplers]"""
import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph
from gfasort_amd import hip, quality as Q
g = G.tile_series(load("DRB1-3123.gfa"), 120)
og = oracle_graph(g)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
print("DRB1 x120 -p Y --iter-max 100: launch mode | G upd/s | stress | d1 rms | d1 trimmed | RMSE", flush=True)
VARIANTS = (("reference streams", hip.F_BUNDLE(1)), ("default (fused, 16 counters)", 0), ("default again", 0), ("one launch per iteration", hip.F_NO_FUSE),
            ("one launch per iteration", hip.F_NO_FUSE), ("free-running", hip.F_DBG_FREE_RUNNING))
if "--samplers" in sys.argv:
    VARIANTS = (("B = 32 (runs of one trip)", hip.F_BUNDLE(32)), ("B = 16 (runs of one trip)", hip.F_BUNDLE(16)), ("B = 8", hip.F_BUNDLE(8)), ("one partner", hip.F_ONE_PARTNER),
                ("no twin trips", hip.F_DBG_NO_TWIN_TRIP), ("no line alignment", hip.F_DBG_NO_ALIGN), ("B = 64, runs of one trip", hip.F_BUNDLE(64) | hip.F_CHAIN(1)))
for name, flags in VARIANTS:
    ctx = hip.Context(g)
    ctx.setup_1d(p, hip.make_config(flags=flags)); ctx.init_positions(); ctx.run()
    x, st = ctx.download(), ctx.stats(); ctx.close()
    sr = Q.short_range_error(g, x, 0, (1,))
    o = hip.sort_order(x).astype(np.int64)
    q = Q.layout_quality(g, o)
    print(f"{name:30s} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f}  {O.stress_1d(og, x, 2_000_000):.5f}  {sr['rms']:.2f}  {sr['trimmed_rms']:.2f}  {q['rmse']:.1f}", flush=True)
