"""C4 (-p L --dimensions 2 and 3): the layout team kernel's rate per run length K and stream count.   python scripts/nd_k_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, sgd as S   # noqa: E402

g = G.synth_windows(1_000_000, 64, 156_250, 2)
ctx = hip.Context(g)
for dims in (2, 3):
    p = P.LayoutSGDParams.from_graph(g, dims, 1)
    p.iter_max = 8
    c0 = S.default_layout_init(g, dims, p.seed)
    for K in (0, 8, 16, 32, 64):
        for T in (0, 131072, 262144):
            if T and K:
                continue
            ctx.setup_nd(p, hip.make_config(n_streams=T, flags=hip.F_CHAIN(K)))
            ctx.upload(c0.ravel())
            ctx.run_iteration(0)
            ctx.synchronize()
            s0 = ctx.stats()
            ctx.run_range(list(range(1, 9)))
            ctx.synchronize()
            s1 = ctx.stats()
            ms = (s1.kernel_ms - s0.kernel_ms) / 8
            print(f"D = {dims}  K {s1.run_trips:2d}  streams {s1.n_streams:6d}: {ms:.3f} ms per iteration = "
                  f"{(s1.term_updates - s0.term_updates) / 8 / (ms * 1e-3) / 1e9:.1f} G updates/s", flush=True)
ctx.close()
