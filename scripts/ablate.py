"""Ablation probe: which memory stream bounds the 1D kernel on C3?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip

def run(g, p, flags, T=0, block=256, n_iter=10, k0=3):
    ctx = hip.Context(g)
    ctx.setup_1d(p, hip.make_config(n_streams=T, flags=flags, block_size=block))
    ctx.upload(hip.init_positions(g))
    for k in range(3):
        ctx.run_iteration(k)
    ctx.synchronize()
    s0 = ctx.stats()
    for k in range(k0, k0 + n_iter):
        ctx.run_iteration(k)
    ctx.synchronize()
    s1 = ctx.stats()
    ctx.close()
    return (s1.term_updates - s0.term_updates) / ((s1.kernel_ms - s0.kernel_ms) * 1e-3) / 1e9

def main():
    shuffle = True
    for (N, Pn, W, label) in [(1_000_000, 64, 156_250, "C3"), (100_000, 1, 100_000, "C2"), (10_000_000, 1024, 97_656, "C5")]:
        g = G.synth_windows(N, Pn, W, 2, shuffle=shuffle)
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = 200
        for name, fl in [("full", 0), ("no_atomics", 0x100), ("no_xloads", 0x200), ("no_atomics+no_xloads", 0x300)]:
            r1 = run(g, p, fl, k0=3)
            r2 = run(g, p, fl, k0=150)
            print(f"{label} {name:24s} noncool {r1:7.3f} G/s   cooling {r2:7.3f} G/s", flush=True)

if __name__ == "__main__":
    main()
