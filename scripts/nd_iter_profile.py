"""C4 (-p L --dimensions 2, default flags): time, updates and lane-slots per ITERATION of the schedule, for the default
kernel and with single features off.  attempts / updates = lane-slots spent per update (1.0 = every lane of every trip
landed a term); the cooling half (sgd.rs:456: every partner a Zipf jump) has more short-jump trips, whose two colours
each keep half the lanes busy.
    python scripts/nd_iter_profile.py [--quick]      (GFS_LIB_PATH selects another build of the library)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, sgd as S   # noqa: E402

g = G.synth_windows(1_000_000, 64, 156_250, 2)
p = P.LayoutSGDParams.from_graph(g, 2, 1)
c0 = S.default_layout_init(g, 2, p.seed)
ctx = hip.Context(g)
QUICK = "--quick" in sys.argv
VARIANTS = (("default", 0), ("no twin trips", hip.F_DBG_NO_TWIN_TRIP), ("no fused short-jump trips", hip.F_DBG_NO_FUSED_TRIP),
            ("one partner", hip.F_ONE_PARTNER), ("no atomic adds", hip.F_DBG_NO_ATOMICS))
if QUICK:
    VARIANTS = (VARIANTS[0], VARIANTS[0], VARIANTS[-1])
for name, flags in VARIANTS:
    ctx.setup_nd(p, hip.make_config(flags=flags))
    ctx.upload(c0)
    rows = []
    for k in range(int(p.iter_max) + 1):
        s0 = ctx.stats()
        ctx.run_iteration(k)
        ctx.synchronize()
        s1 = ctx.stats()
        rows.append((k, s1.kernel_ms - s0.kernel_ms, s1.term_updates - s0.term_updates, s1.attempts - s0.attempts))
    tot_ms = sum(r[1] for r in rows[1:])
    tot_u = sum(r[2] for r in rows[1:])
    print(f"{name}: {tot_u / (tot_ms * 1e-3) / 1e9:.1f} G updates/s over iterations 1..{int(p.iter_max)}  ({s1.n_streams} streams, "
          f"runs of {s1.run_trips} trips)", flush=True)
    for k, ms, u, a in rows:
        if QUICK:
            continue
        if name == "default" or k in (1, 8, 14, 16, 20, 30):
            print(f"    iteration {k:2d}: {ms:.3f} ms  {u / (ms * 1e-3) / 1e9:5.1f} G updates/s  lane-slots per update {a / max(u, 1):.3f}", flush=True)
ctx.close()
