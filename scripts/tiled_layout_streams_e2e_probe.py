import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph
from gfasort_amd import hip, quality as Q, sgd as S
g = G.tile_series(load("DRB1-3123.gfa"), 120)
og = oracle_graph(g)
print("DRB1 x120 -p L --dimensions 2 --layout-iter 90: streams | G upd/s | stress | rel. error d1, 2-3, 64-127, 256-511 | e2e median, mean", flush=True)
for T in (65536, 131072, 196608):
    acc = []
    for seed in range(4):
        p = P.LayoutSGDParams.from_graph(g, 2, 1); p.iter_max = 90; p.seed = p.seed + 1000 * seed
        c0 = S.default_layout_init(g, 2, p.seed)
        rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(n_streams=T))
        _, rms, _ = Q.stress_by_scale(g, c, 2, 1_000_000)
        cc = np.asarray(c).reshape(-1, 2, 2)
        err = np.abs(np.sqrt(((cc[:, 0, :] - cc[:, 1, :]) ** 2).sum(axis=1)) - g.node_len)
        acc.append([st.term_updates / (st.kernel_ms * 1e-3) / 1e9, O.layout_stress(og, 2, c, 2_000_000), rms[0], rms[1], rms[6], rms[8], float(np.median(err)), float(np.mean(err))])
    print(f"{T:7d}  " + " ".join(f"{v:.4f}" for v in np.mean(acc, axis=0)) + "   (stress per seed: " + " ".join(f"{r[1]:.4f}" for r in acc) + ")", flush=True)
