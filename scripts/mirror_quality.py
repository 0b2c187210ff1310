"""The oracle's sequential mirror of the team sampler on the 525k-node bubble graph (CPU, ~30 s per run): what the sampler's
structure costs with NO concurrency — one partner per leader against two.  usage: mirror_quality.py [n_streams]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from util import O, G, P, oracle_graph, oracle_params
from gfasort_amd import quality as Q
from gfasort_amd.distributed import path_order_layout

g = G.synth_bubbles(400_000, 24, 6)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
og, op = oracle_graph(g), oracle_params(p)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
for mode, kw in (("p1", dict(partners=1)), ("twin", dict(partners=2))):
    t0 = time.time()
    st = O.State(og, op, n_streams=T, bundle=64, node_slots=path_order_layout(g), chain=64, **kw)
    x = O.init_positions(og)
    st.run(x)
    _, rms, cnt = Q.stress_by_scale(g, x, 0, 1_000_000)
    print(mode, "T", T, "stress", O.stress_1d(og, x, 2_000_000), "octaves", np.round(rms, 4).tolist(), f"({time.time()-t0:.0f} s)", flush=True)
