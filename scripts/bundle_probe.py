"""Experiment: throughput and quality of bundled sampling (GFS_F_BUNDLE) vs reference streams."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip
from oracle import oracle as O

def quality(g, x):
    og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
    st = O.stress_1d(og, x, 200000)
    ids = g.node_ids[np.argsort(x, kind="stable")].astype(np.int64)
    fwd = ids if ids[0] < ids[-1] else ids[::-1]
    inv = int((np.diff(fwd) < 0).sum())
    return st, inv

def run(g, p, bundle, T=0):
    ctx = hip.Context(g)
    ctx.setup_1d(p, hip.make_config(n_streams=T, flags=(bundle << 16)))
    ctx.upload(hip.init_positions(g))
    t0 = time.time()
    ctx.run()
    dt = time.time() - t0
    st = ctx.stats()
    x = ctx.download()
    ctx.close()
    return x, st, dt

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "c3"):
        g = G.synth_windows(1_000_000, 64, 156_250, 2)
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = 200
        for b in (1, 4, 8, 16, 32, 64):
            x, st, dt = run(g, p, b)
            s, inv = quality(g, x)
            print(f"C3 bundle={b:2d}: {st.term_updates/ (st.kernel_ms*1e-3)/1e9:7.3f} G upd/s (kernel) wall {dt:.3f}s updates {st.term_updates} "
                  f"attempts/upd {st.attempts/st.term_updates:.4f} stress {s:.3e} adjacent inversions {inv}", flush=True)
    if which in ("all", "c2"):
        g = G.synth_chain(100_000, 1)
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        for b in (1, 8, 16, 64):
            x, st, dt = run(g, p, b)
            s, inv = quality(g, x)
            print(f"C2 bundle={b:2d}: {st.term_updates/ (st.kernel_ms*1e-3)/1e9:7.3f} G upd/s streams {st.n_streams} stress {s:.3e} inversions {inv}", flush=True)
    if which in ("all", "drb1"):
        g = G.load_gfa(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "data", "DRB1-3123.gfa"))
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
        print("DRB1 initial stress", O.stress_1d(og, O.init_positions(og), 200000))
        for b in (1, 4, 8, 16, 32, 64):
            res = []
            for seed in range(3):
                p.seed = 9399220 + 1000 * seed
                x, st, dt = run(g, p, b)
                res.append(O.stress_1d(og, x, 200000))
            print(f"DRB1 bundle={b:2d}: streams {st.n_streams} stress over 3 seeds {['%.4f' % v for v in res]}", flush=True)
        p.seed = 9399220
        xr = O.init_positions(og)
        kw = {k: getattr(p, k) for k in ["iter_max", "iter_with_max_learning_rate", "min_term_updates", "delta", "eps", "eta_max", "theta", "space", "space_max", "space_quantization_step", "cooling_start", "seed"]}
        O.sgd_1d(og, O.params(**kw), xr, n_streams=8)
        print("DRB1 oracle 8 streams stress", O.stress_1d(og, xr, 200000))

if __name__ == "__main__":
    main()
