"""Where does the spread of the distance-1 figure come from?  (tests/test_gpu_quality.py: pools at 209 920 streams once gave
0.2048 against 0.191 typical.)  525k-node bubble graph, default `-p Y`: for the default stream count and for 209 920
streams, work pools against free-running waves, N_SEEDS RNG seeds x 2 repeats of each seed (same seed twice = the
concurrency's own run-to-run spread; different seeds = the sampler's), and the instrument's own noise (the same positions
measured with 3 sample seeds).  Reference streams (3 seeds) for scale.
    python scripts/pool_seed_study.py [n_seeds = 5]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, quality as Q   # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = G.synth_bubbles(400_000, 24, 6)
ctx = hip.Context(g)


def run(seed, n_streams, flags):
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.seed = seed
    ctx.setup_1d(p, hip.make_config(n_streams=n_streams, flags=flags))
    ctx.init_positions()
    ctx.run()
    st = ctx.stats()
    assert st.term_updates == (p.iter_max + 1) * p.min_term_updates
    return ctx.download(), st


def d(x, seed=777):
    _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000, seed=seed)
    return rms[:4]


print("525k-node bubble graph, -p Y --iter-max 100; relative error at path distance 1 | 2-3 | 4-7 | 8-15", flush=True)
x, st = run(9399220, 0, 0)
print(f"instrument noise (same positions, 3 sample seeds): " + "  ".join(" ".join(f"{v:.4f}" for v in d(x, s)) for s in (777, 778, 779)), flush=True)
print(f"default stream count = {st.n_streams}", flush=True)
for label, n_streams, flags in (("default count, pools", 0, 0), ("default count, free-running", 0, hip.F_DBG_FREE_RUNNING),
                                ("209920 streams, pools", 209_920, 0), ("209920 streams, free-running", 209_920, hip.F_DBG_FREE_RUNNING),
                                ("131072 streams, pools", 131_072, 0)):
    vals = []
    for s in range(n_seeds):
        for rep in range(2):
            x, st = run(9399220 + 1000 * s, n_streams, flags)
            vals.append(d(x))
    a = np.array(vals)                                    # [seed*2+rep, octave]
    d1 = a[:, 0].reshape(n_seeds, 2)
    print(f"{label:30s} d1 by seed (2 repeats each): " + "  ".join(f"{u:.4f}/{v:.4f}" for u, v in d1), flush=True)
    print(f"{'':30s} mean {a.mean(axis=0).round(4).tolist()}  sd {a.std(axis=0).round(4).tolist()}  "
          f"d1: sd between seed means {d1.mean(axis=1).std():.4f}, mean |repeat difference| {np.abs(d1[:, 0] - d1[:, 1]).mean():.4f}, "
          f"max {a[:, 0].max():.4f}", flush=True)
vals = [d(run(9399220 + 1000 * s, 0, hip.F_BUNDLE(1))[0]) for s in range(3)]
a = np.array(vals)
print(f"{'reference streams (B = 1)':30s} d1 by seed: " + "  ".join(f"{v:.4f}" for v in a[:, 0]) +
      f"   mean {a.mean(axis=0).round(4).tolist()}", flush=True)
ctx.close()
