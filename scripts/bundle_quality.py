"""Quality of bundled sampling vs reference streams on bubble graphs and DRB1 (stress at equal update counts)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from oracle import oracle as O

def run(g, p, bundle, T=0):
    ctx = hip.Context(g)
    ctx.setup_1d(p, hip.make_config(n_streams=T, flags=(bundle << 16)))
    ctx.upload(hip.init_positions(g))
    ctx.run()
    st = ctx.stats()
    x = ctx.download()
    ctx.close()
    return x, st

def study(name, g, iters_list, bundles, seeds=3, T=0):
    og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    print(f"== {name}: nodes {g.n_nodes} steps {g.n_steps} paths {g.n_paths} M {p.min_term_updates} initial stress {O.stress_1d(og, O.init_positions(og), 200000):.4f}", flush=True)
    for iters in iters_list:
        p.iter_max = iters
        for b in bundles:
            res, rate = [], 0
            for sd in range(seeds):
                p.seed = 9399220 + 1000 * sd
                x, st = run(g, p, b, T)
                res.append(O.stress_1d(og, x, 200000))
                rate = st.term_updates / (st.kernel_ms * 1e-3) / 1e9
            print(f"  iter_max {iters:4d} bundle {b:2d}: stress mean {np.mean(res):.4f} [{' '.join('%.4f' % v for v in res)}]  {rate:.2f} G/s streams {st.n_streams}", flush=True)

def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = G.load_gfa(os.path.join(root, "tests", "data", "DRB1-3123.gfa"))
    study("DRB1", g, [100, 300], [1, 8, 16, 64])
    study("DRB1 T=512", g, [100], [1, 8, 16, 64], T=512)
    study("bubbles 20k sites x 16 hap", G.synth_bubbles(20000, 16, 5), [100], [1, 8, 16, 64])
    study("bubbles 400k sites x 24 hap", G.synth_bubbles(400000, 24, 6), [100], [1, 8, 16, 64], seeds=2)

if __name__ == "__main__":
    main()
