"""What one rank of an 8-GPU run sees: C3 with 1/8 of the updates per iteration, fused windows of 8 iterations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip


def main():
    g = G.synth_windows(1_000_000, 64, 156_250, 2)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = 200
    ctx = hip.Context(g)
    x0 = hip.init_positions(g)
    for div in (8, 4, 2):
        quota = int(p.min_term_updates) // div
        for B in (64, 32, 16):
            for T in (0, 65536, 32768, 16384):
                ctx.setup_1d(p, hip.make_config(n_streams=T, term_updates_per_iteration=quota, flags=hip.F_BUNDLE(B)))
                ctx.upload(x0)
                ctx.run_range(list(range(8)))
                ctx.synchronize()
                s0 = ctx.stats()
                for w in range(1, 25):
                    ctx.run_range(list(range(8 * w, 8 * w + 8)))
                ctx.synchronize()
                s1 = ctx.stats()
                ms = (s1.kernel_ms - s0.kernel_ms) / 24
                upd = (s1.term_updates - s0.term_updates) / 24
                print(f"quota 1/{div} B={B:2d} T={s1.n_streams:6d}: window of 8 its {ms * 1e3:7.1f} us  {upd / ms / 1e6:6.2f} G upd/s",
                      flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
