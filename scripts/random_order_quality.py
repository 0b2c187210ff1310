"""Unsorted input (what gfasort exists for): nodes in random input order, so the SGD starts from a random
arrangement.  Sampled stress at equal update counts: CPU oracle (8 reference streams) vs the GPU kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from oracle import oracle as O
from order_probe_util import permuted

def main():
    for name, g0, iters in (("windows 30k nodes x 8 paths", G.synth_windows(30_000, 8, 15_000, 4, shuffle=False), 100),
                            ("bubbles 20k sites x 16 hap", G.synth_bubbles(20000, 16, 5), 100)):
        g = permuted(g0, np.random.default_rng(1).permutation(g0.n_nodes))
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = iters
        og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
        op = O.params(**{k: getattr(p, k) for k in ["iter_max", "iter_with_max_learning_rate", "min_term_updates", "delta", "eps",
                                                      "eta_max", "theta", "space", "space_max", "space_quantization_step",
                                                      "cooling_start", "seed"]})
        x0 = O.init_positions(og)
        s0 = O.stress_1d(og, x0, 200000)
        xo = x0.copy()
        O.sgd_1d(og, op, xo, n_streams=8)
        print(f"== {name}, random input order: initial stress {s0:.4g}; oracle 8 streams {O.stress_1d(og, xo, 200000):.4g}", flush=True)
        for B in (1, 0, 64):
            rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(flags=hip.F_BUNDLE(B)))
            print(f"   GPU bundle req {B:2d} used {st.bundle:2d} streams {st.n_streams:6d}: stress {O.stress_1d(og, x, 200000):.4g}", flush=True)

if __name__ == "__main__":
    main()
