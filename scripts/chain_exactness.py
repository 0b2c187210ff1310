"""Inversions left in the final order of small chains (the hardest exactness cases), several seeds, for sampler
variants selected by debug flags: 0 = product, 0x2000 = first-run-only alignment, 0x1000 = no alignment."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip

def inversions(g, x):
    ids = g.node_ids[np.argsort(x, kind="stable")].astype(np.int64)
    d = np.diff(ids)
    fwd = (d != 1).sum(); bwd = (d != -1).sum()
    return int(min(fwd, bwd))

def main():
    for name, g in (("chain 20k x 1 path", G.synth_chain(20000, 1)), ("chain 20k x 4 paths", G.synth_chain(20000, 4)),
                    ("chain 100k x 1 path", G.synth_chain(100000, 1)), ("windows 50k x 8 paths", G.synth_windows(50_000, 8, 25_000, 6))):
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        for flags, label in ((0, "product"), (0x2000, "first run only"), (0x1000, "no alignment")):
            res = []
            for sd in range(8):
                p.seed = 9399220 + 1000 * sd
                rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(flags=flags))
                res.append(inversions(g, x))
            print(f"{name:24s} bundle {st.bundle:2d} {label:16s}: inversions per seed {res}", flush=True)

if __name__ == "__main__":
    main()
