"""HIP-event time of a fused launch (gfs_stats.kernel_ms) for the first and the following launches of a process:
is the first dispatch of the fused kernel slower than the kernel itself?  (run under rocprofv3 --kernel-trace to compare)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip

g = G.synth_windows(1_000_000, 64, 156_250, 2)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
p.iter_max = 200
ctx = hip.Context(g)
ctx.setup_1d(p, hip.make_config())
ctx.init_positions()
for k in range(5):
    ctx.run_iteration(k)
ctx.synchronize()
prev = ctx.stats().kernel_ms
for rep in range(4):
    t0 = time.perf_counter()
    ctx.run_range(list(range(20)))
    ctx.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    now = ctx.stats().kernel_ms
    print(f"fused launch {rep}: events {now - prev:.4f} ms, wall {wall:.4f} ms", flush=True)
    prev = now
