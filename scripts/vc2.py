import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from virtual_cluster import *
g = G.synth_bubbles(400000, 24, 6)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
for seed in (9399220, 77):
    p.seed = seed
    for sharding in ("contiguous", "lpt"):
        for K in (1, 8):
            x = run_cluster(g, p, 8, merge_every=K, sharding=sharding)
            print(f"seed {seed} {sharding:10s} every {K}: {quality(g, x, False)}", flush=True)
g = G.load_gfa(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "data", "DRB1-3123.gfa"))
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
rc, x1, st = hip.path_linear_sgd_raw(g, p)
print("DRB1 single", quality(g, x1, False))
for world in (2, 8):
    for K in (1, 8):
        x = run_cluster(g, p, world, merge_every=K, sharding="lpt")
        print(f"DRB1 world {world} every {K}: {quality(g, x, False)}", flush=True)
