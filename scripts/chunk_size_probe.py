"""C3 (-p Y) and C4 (-p L): the fused pooled launch per chunk size (GFS_DBG_CHUNK), 20 iterations and the whole schedule.
    python scripts/chunk_size_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, sgd as S   # noqa: E402

g = G.synth_windows(1_000_000, 64, 156_250, 2)
for dims in (0, 2):
    p = P.LayoutSGDParams.from_graph(g, dims, 1) if dims else P.YgsParams.from_graph(g, 0, 1).path_sgd
    if not dims:
        p.iter_max = 200
    ctx = hip.Context(g)
    ctx.setup_nd(p, hip.make_config()) if dims else ctx.setup_1d(p, hip.make_config())
    x0 = S.default_layout_init(g, dims, p.seed).ravel() if dims else hip.init_positions(g)
    for n in (20, int(p.iter_max)):
        for ch in (512, 1024, 2048, 4096, 8192):
            os.environ["GFS_DBG_CHUNK"] = str(ch)
            best = 1e9
            for rep in range(3):
                ctx.upload(x0)
                ctx.reset_streams()
                ctx.run_iteration(0)
                ctx.synchronize()
                s0 = ctx.stats()
                ctx.run_range(list(range(1, n + 1)))
                ctx.synchronize()
                s1 = ctx.stats()
                best = min(best, s1.kernel_ms - s0.kernel_ms)
            print(f"D = {dims}  {n:3d} iterations in one launch, chunks of {ch:5d}: {best:8.3f} ms = "
                  f"{(s1.term_updates - s0.term_updates) / (best * 1e-3) / 1e9:6.1f} G updates/s", flush=True)
    ctx.close()
