"""C4 (-p L --dimensions 2): the layout team kernel built for 2 (default) or 3 waves per SIMD (GFS_LIB_PATH selects the
experiment build), per stream count.   python scripts/nd_waves_probe.py"""
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
print("library:", hip.lib_path() if not os.environ.get("GFS_LIB_PATH") else os.environ["GFS_LIB_PATH"], flush=True)
for T in (131072, 163840, 196608):
    for flags, name in ((0, "default"), (hip.F_CHAIN(64), "K = 64")):
        ctx.setup_nd(p, hip.make_config(n_streams=T, flags=flags))
        ctx.upload(c0)
        ctx.run_iteration(0)
        ctx.synchronize()
        s0 = ctx.stats()
        ctx.run_range([1, 2, 3, 4])
        ctx.synchronize()
        s1 = ctx.stats()
        ms = (s1.kernel_ms - s0.kernel_ms) / 4
        print(f"{name:10s} T {s1.n_streams:6d} K {s1.run_trips:2d}: {ms:.3f} ms per iteration of 1e8 updates = "
              f"{(s1.term_updates - s0.term_updates) / 4 / (ms * 1e-3) / 1e9:.1f} G updates/s", flush=True)
ctx.close()
