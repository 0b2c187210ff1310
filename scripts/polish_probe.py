"""Prototype of a cooling bundle schedule: B = 64 for the first part of the run, a narrow bundle for the last part."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from oracle import oracle as O

g = G.synth_bubbles(1_500_000, 32, 7)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
x0 = hip.init_positions(g)
ctx = hip.Context(g)
n = int(p.iter_max) + 1
# head variants: a narrow bundle FIRST (while eta is large), B = 64 afterwards
for head_frac, head_B in ((0.1, 1), (0.2, 1), (0.1, 8), (0.3, 8), (0.5, 8)):
    k1 = int(round(head_frac * n))
    ctx.setup_1d(p, hip.make_config(flags=hip.F_BUNDLE(head_B)))
    ctx.upload(x0)
    ctx.run_range(list(range(0, k1))); ctx.synchronize()
    ms = ctx.stats().kernel_ms
    x = ctx.download()
    ctx.setup_1d(p, hip.make_config(flags=hip.F_BUNDLE(64), stream_base=1 << 22))
    ctx.upload(x)
    ctx.run_range(list(range(k1, n))); ctx.synchronize()
    ms += ctx.stats().kernel_ms
    x = ctx.download()
    print(f"bubbles 1.5M: B={head_B} for {k1} iterations, then B=64 for {n - k1}: kernels {ms:7.1f} ms  stress {O.stress_1d(og, x, 200000):.4g}", flush=True)
for tail_frac, tail_B in ((0.0, 64),):
    k1 = n - int(round(tail_frac * n))
    ctx.setup_1d(p, hip.make_config(flags=hip.F_BUNDLE(64)))
    ctx.upload(x0)
    ctx.run_range(list(range(0, k1))); ctx.synchronize()
    ms = ctx.stats().kernel_ms
    x = ctx.download()
    if k1 < n:
        ctx.setup_1d(p, hip.make_config(flags=hip.F_BUNDLE(tail_B), stream_base=1 << 22))
        ctx.upload(x)
        ctx.run_range(list(range(k1, n))); ctx.synchronize()
        ms += ctx.stats().kernel_ms
        x = ctx.download()
    print(f"bubbles 1.5M: B=64 for {k1} iterations, then B={tail_B} for {n - k1}: kernels {ms:7.1f} ms  stress {O.stress_1d(og, x, 200000):.4g}", flush=True)
ctx.close()
