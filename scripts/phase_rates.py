"""C3: rate of the team kernel in the two halves of the schedule (zipf theta=0.99 + uniform far jumps, then cooling)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfasort_amd import graph as G, params as P, hip

g = G.synth_windows(1_000_000, 64, 156_250, 2)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
p.iter_max = 200
ctx = hip.Context(g)
ctx.setup_1d(p, hip.make_config())
ctx.upload(hip.init_positions(g))
ctx.run_range(list(range(0, 5))); ctx.synchronize()
for name, ks in (("iterations 5..100 (not cooling)", range(5, 101)), ("iterations 101..200 (cooling)", range(101, 201))):
    s0 = ctx.stats()
    ctx.run_range(list(ks)); ctx.synchronize()
    s1 = ctx.stats()
    ms = s1.kernel_ms - s0.kernel_ms
    upd = s1.term_updates - s0.term_updates
    print(f"{name}: {upd / ms / 1e6:6.2f} G updates/s, attempts per update {(s1.attempts - s0.attempts) * 1.0 / upd:.3f}", flush=True)
ctx.close()
