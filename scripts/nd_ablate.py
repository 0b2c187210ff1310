"""C4 (-p L --dimensions 2), the default layout kernel: what its time is made of.  Rate with everything, without the atomic
adds, without the coordinate loads of twin trips (both ablations give wrong results), with one partner, per stream count.
    python scripts/nd_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, sgd as S   # noqa: E402

g = G.synth_windows(1_000_000, 64, 156_250, 2)
p = P.LayoutSGDParams.from_graph(g, 2, 1)
p.iter_max = 5
c0 = S.default_layout_init(g, 2, p.seed)
ctx = hip.Context(g)
import os as _os
for name, flags, T in (("default", 0, 0), ("adds regrouped but not issued (GFS_DBG2=1)", -1, 0), ("no atomic adds", hip.F_DBG_NO_ATOMICS, 0), ("no coordinate loads (twin trips)", hip.F_DBG_NO_XLOADS, 0),
                       ("neither", hip.F_DBG_NO_ATOMICS | hip.F_DBG_NO_XLOADS, 0), ("one partner", hip.F_ONE_PARTNER, 0),
                       ("no twin trips", hip.F_DBG_NO_TWIN_TRIP, 0), ("default, 65536 streams", 0, 65536), ("default, 98304 streams", 0, 98304)):
    _os.environ.pop("GFS_DBG2", None)
    if flags == -1:
        _os.environ["GFS_DBG2"] = "1"
        flags = 0
    ctx.setup_nd(p, hip.make_config(n_streams=T, flags=flags))
    ctx.upload(c0)
    ctx.run_iteration(0)
    ctx.synchronize()
    s0 = ctx.stats()
    ctx.run_range([1, 2, 3, 4])
    ctx.synchronize()
    s1 = ctx.stats()
    ms = (s1.kernel_ms - s0.kernel_ms) / 4
    print(f"{name:44s} T {s1.n_streams:6d} K {s1.run_trips:2d}: {ms:.3f} ms per iteration of 1e8 updates = "
          f"{(s1.term_updates - s0.term_updates) / 4 / (ms * 1e-3) / 1e9:.1f} G updates/s", flush=True)
ctx.close()
