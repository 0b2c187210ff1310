"""Quick throughput probe on the C3 graph (1M nodes / 64 paths / 10M steps)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip

def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    t0 = time.time()
    g = G.synth_windows(1_000_000, 64, 156_250, 2)
    print("graph built", time.time() - t0, g.n_nodes, g.n_steps, flush=True)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = 200
    print("params", p, flush=True)
    for flags, block, T in [(0, 256, 0), (hip.F_PLAIN_LOADS, 256, 0), (0, 512, 0), (0, 256, 262144), (0, 256, 1048576), (hip.F_NO_LDS_TABLES, 256, 0)]:
        ctx = hip.Context(g)
        cfg = hip.make_config(n_streams=T, flags=flags, block_size=block)
        t0 = time.time()
        ctx.setup_1d(p, cfg)
        ctx.upload(hip.init_positions(g))
        print("setup", time.time() - t0, flush=True)
        for k in range(3):
            ctx.run_iteration(k)
        ctx.synchronize()
        s0 = ctx.stats()
        t0 = time.time()
        for k in range(3, 3 + n_iter):
            ctx.run_iteration(k)
        ctx.synchronize()
        dt = time.time() - t0
        s1 = ctx.stats()
        upd = s1.term_updates - s0.term_updates
        kms = s1.kernel_ms - s0.kernel_ms
        print(f"flags={flags} block={block} T={s1.n_streams}: {upd} updates wall {dt*1e3:.2f} ms kernel {kms:.2f} ms "
              f"-> {upd/dt/1e9:.3f} G upd/s wall, {upd/(kms*1e-3)/1e9:.3f} G upd/s kernel; attempts/updates={(s1.attempts-s0.attempts)/upd:.4f}", flush=True)
        # cooling-phase iterations
        t0 = time.time()
        for k in range(150, 150 + n_iter):
            ctx.run_iteration(k)
        ctx.synchronize()
        dt = time.time() - t0
        s2 = ctx.stats()
        upd = s2.term_updates - s1.term_updates
        kms = s2.kernel_ms - s1.kernel_ms
        print(f"   cooling: {upd/dt/1e9:.3f} G upd/s wall, {upd/(kms*1e-3)/1e9:.3f} G upd/s kernel", flush=True)
        ctx.close()

if __name__ == "__main__":
    main()
