import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gfasort_amd import graph as G, params as P, hip, sgd as S
g = G.synth_windows(1_000_000, 64, 156_250, 2)
ctx = hip.Context(g)
p = P.LayoutSGDParams.from_graph(g, 2, 1)
p.iter_max = 8
c0 = S.default_layout_init(g, 2, p.seed)
for K in (16, 64):
    for T in (196608, 393216, 589824, 786432, 1179648):
        try:
            ctx.setup_nd(p, hip.make_config(n_streams=T, flags=hip.F_CHAIN(K)))
        except Exception as e:
            print(K, T, "refused:", e); continue
        ctx.upload(c0.ravel()); ctx.run_iteration(0); ctx.synchronize()
        s0 = ctx.stats(); ctx.run_range(list(range(1, 9))); ctx.synchronize(); s1 = ctx.stats()
        ms = (s1.kernel_ms - s0.kernel_ms) / 8
        print(f"K {s1.run_trips:2d} streams {s1.n_streams:7d}: {ms:.3f} ms = {(s1.term_updates - s0.term_updates) / 8 / (ms * 1e-3) / 1e9:.1f} G updates/s", flush=True)
ctx.close()
