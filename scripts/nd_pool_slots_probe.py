import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph
from gfasort_amd import hip, quality as Q, sgd as S
g = G.tile_series(load("DRB1-3123.gfa"), 120)
og = oracle_graph(g)
for name, env in (("16 slots", "0"), ("one slot", "256")):
    os.environ["GFS_DBG2"] = env
    for seed in range(2):
        p = P.LayoutSGDParams.from_graph(g, 2, 1); p.iter_max = 90; p.seed = p.seed + 1000 * seed
        c0 = S.default_layout_init(g, 2, p.seed)
        rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0)
        cc = np.asarray(c).reshape(-1, 2, 2)
        err = np.abs(np.sqrt(((cc[:, 0, :] - cc[:, 1, :]) ** 2).sum(axis=1)) - g.node_len)
        print(name, seed, st.launches, f"{st.term_updates / (st.kernel_ms * 1e-3) / 1e9:.1f} G/s", f"stress {O.layout_stress(og, 2, c, 2_000_000):.4f} e2e median {np.median(err):.3f} mean {np.mean(err):.3f}", flush=True)
# C4 rate with one slot
g4 = G.synth_windows(1_000_000, 64, 156_250, 2)
p = P.LayoutSGDParams.from_graph(g4, 2, 1)
c0 = S.default_layout_init(g4, 2, p.seed)
for env in ("0", "256"):
    os.environ["GFS_DBG2"] = env
    rc, c, st = hip.path_linear_sgd_layout_raw(g4, p, c0)
    print("C4 GFS_DBG2", env, st.launches, f"{st.term_updates / (st.kernel_ms * 1e-3) / 1e9:.1f} G/s", flush=True)
