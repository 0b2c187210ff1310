"""Tandem repeats: paths that step on the same node (or the same short cycle of nodes) many times in a row, so that
a run of 64 consecutive steps hits a node repeatedly.  Stress after the default run: oracle vs GPU modes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from oracle import oracle as O

repeats_graph = G.synth_repeats

def main():
    for name, g in (("self-loops x<=40 every 500 nodes", repeats_graph(100_000, 16, 1, 40, 500, 1)),
                    ("2-cycles x<=30 every 500 nodes", repeats_graph(100_000, 16, 2, 30, 500, 2)),
                    ("5-cycles x<=20 every 300 nodes", repeats_graph(100_000, 16, 5, 20, 300, 3)),
                    ("self-loops x<=200 every 2000 nodes", repeats_graph(100_000, 16, 1, 200, 2000, 4))):
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
        op = O.params(**{k: getattr(p, k) for k in ["iter_max", "iter_with_max_learning_rate", "min_term_updates", "delta", "eps",
                                                      "eta_max", "theta", "space", "space_max", "space_quantization_step",
                                                      "cooling_start", "seed"]})
        xo = O.init_positions(og)
        s0 = O.stress_1d(og, xo, 200000)
        O.sgd_1d(og, op, xo, n_streams=8)
        line = f"== {name}: steps {g.n_steps} initial {s0:.4g} oracle {O.stress_1d(og, xo, 200000):.4g} |"
        for B in (1, 16, 64):
            rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(flags=hip.F_BUNDLE(B)))
            line += f" GPU B={st.bundle}: {O.stress_1d(og, x, 200000):.4g} (finite {bool(np.isfinite(x).all())})"
        print(line, flush=True)

if __name__ == "__main__":
    main()
