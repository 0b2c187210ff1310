"""Seeds study: random input order, medium windows graphs; stress after the default 100 iterations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from oracle import oracle as O
from order_probe_util import permuted

def main():
    for name, g0 in (("windows 30k nodes x 8 paths", G.synth_windows(30_000, 8, 15_000, 4, shuffle=False)),
                     ("windows 100k nodes x 32 paths", G.synth_windows(100_000, 32, 31_250, 2, shuffle=False))):
        g = permuted(g0, np.random.default_rng(1).permutation(g0.n_nodes))
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
        print(f"== {name}, random input order, M {p.min_term_updates}", flush=True)
        res = []
        for sd in range(3):
            p.seed = 9399220 + 1000 * sd
            op = O.params(**{k: getattr(p, k) for k in ["iter_max", "iter_with_max_learning_rate", "min_term_updates", "delta", "eps",
                                                          "eta_max", "theta", "space", "space_max", "space_quantization_step",
                                                          "cooling_start", "seed"]})
            xo = O.init_positions(og)
            O.sgd_1d(og, op, xo, n_streams=8)
            res.append(O.stress_1d(og, xo, 200000))
        print("   oracle 8 streams      : " + " ".join(f"{v:.3g}" for v in res), flush=True)
        for B in (1, 8, 16, 32, 64):
            res = []
            for sd in range(5):
                p.seed = 9399220 + 1000 * sd
                rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(flags=hip.F_BUNDLE(B)))
                res.append(O.stress_1d(og, x, 200000))
            print(f"   GPU bundle {st.bundle:2d} streams {st.n_streams:6d}: " + " ".join(f"{v:.3g}" for v in res) + f"   median {np.median(res):.3g}", flush=True)

if __name__ == "__main__":
    main()
