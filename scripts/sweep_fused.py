"""Fused persistent launch: whole -p Y run of C3 against streams per launch and block size; checks the order."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip


def chain_ok(g, x):
    order = np.argsort(x, kind="stable")
    ids = g.node_ids[order]
    return bool(np.all(np.diff(ids) == 1) or np.all(np.diff(ids) == -1))


def main():
    g = G.synth_windows(1_000_000, 64, 156_250, 2)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = 200
    ctx = hip.Context(g)
    x0 = hip.init_positions(g)
    for B in (64, 32):
        for block in (256, 64, 128, 512):
            for T in (65536, 131072, 196608, 262144, 393216, 524288):
                if block != 256 and T not in (131072, 262144):
                    continue
                ctx.setup_1d(p, hip.make_config(n_streams=T, flags=hip.F_BUNDLE(B), block_size=block))
                ctx.upload(x0)
                s0 = ctx.stats()
                ctx.run()
                s1 = ctx.stats()
                ms = s1.kernel_ms - s0.kernel_ms
                upd = s1.term_updates - s0.term_updates
                x = ctx.download()
                print(f"B={B} block={block:3d} T={T:7d}: {upd / ms / 1e6:7.2f} G upd/s  kernel {ms:7.2f} ms  "
                      f"order_ok={chain_ok(g, x)}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
