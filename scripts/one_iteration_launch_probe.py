"""C3 (-p Y) and C4 (-p L): ONE iteration per launch — fixed quotas per wave (gfs_ctx_run_iteration) against the pooled launch with
shortened chunks (gfs_ctx_run_range with one iteration), and the fused launch of the same iterations for scale.
    python scripts/one_iteration_launch_probe.py"""
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
    ks = list(range(1, 21))
    variants = [("fixed quotas, one launch per iteration", None, lambda: [ctx.run_iteration(k) for k in ks]),
                ("pooled, one launch per iteration", None, lambda: [ctx.run_range([k]) for k in ks])]
    for ch in (256, 512, 1024, 2048):                                   # GFS_DBG_ONE_CHUNK: the chunk of a pooled launch of one iteration
        variants.append((f"pooled, one launch per iteration, chunks of {ch}", str(ch), lambda: [ctx.run_range([k]) for k in ks]))
    variants.append(("pooled, one fused launch", None, lambda: ctx.run_range(ks)))
    for name, env, fn in variants:
        os.environ.pop("GFS_DBG_ONE_CHUNK", None)
        if env:
            os.environ["GFS_DBG_ONE_CHUNK"] = env
        ctx.upload(x0)
        ctx.run_iteration(0)
        ctx.synchronize()
        s0 = ctx.stats()
        fn()
        ctx.synchronize()
        s1 = ctx.stats()
        ms = (s1.kernel_ms - s0.kernel_ms) / len(ks)
        print(f"D = {dims}  {name:56s}: {s1.launches - s0.launches:2d} launches, {ms * 1e3:8.1f} us of kernel per iteration = "
              f"{(s1.term_updates - s0.term_updates) / len(ks) / (ms * 1e-3) / 1e9:6.1f} G updates/s", flush=True)
    ctx.close()
