"""Reference streams in one fused pooled launch (K1d) against one launch per iteration, on the reference's DRB1 fixture at
the CLI's defaults, over several RNG seeds: final sampled stress (200k pairs) and the relative error at path distance 1, 2-3
(gfasort_amd/quality.py), kernel time.  GFS_DBG_REF_CHUNK varies the updates a lane takes per pool claim.
    python scripts/ref_fused_probe.py [n_seeds = 6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph, oracle_params   # noqa: E402
from gfasort_amd import hip, quality as Q   # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = load("DRB1-3123.gfa")
og = oracle_graph(g)
variants = [("unfused", hip.F_NO_FUSE, None)] + [(f"fused chunk {c}", 0, c) for c in (1, 2, 8, 32, 128)]
rows = {}
for name, flags, chunk in variants:
    if chunk is None:
        os.environ.pop("GFS_DBG_REF_CHUNK", None)
    else:
        os.environ["GFS_DBG_REF_CHUNK"] = str(chunk)
    for s in range(n_seeds):
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.seed = 9399220 + 1000 * s
        rc, x, st = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(flags=flags))
        assert rc == 0 and st.term_updates == 101 * p.min_term_updates
        _, rms, _ = Q.stress_by_scale(g, x, 0, 400_000)
        rows.setdefault(name, []).append((O.stress_1d(og, x, 200_000), rms[0], rms[1], st.kernel_ms, st.launches))
for s in range(n_seeds):
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.seed = 9399220 + 1000 * s
    x = O.init_positions(og)
    O.sgd_1d(og, oracle_params(p), x, n_streams=64)
    _, rms, _ = Q.stress_by_scale(g, x, 0, 400_000)
    rows.setdefault("oracle 64 streams", []).append((O.stress_1d(og, x, 200_000), rms[0], rms[1], 0.0, 0))
print(f"DRB1-3123, -p Y --iter-max 100, {n_seeds} seeds: mean (sd)")
for name, r in rows.items():
    a = np.array(r)
    print(f"{name:20s} stress {a[:, 0].mean():.4f} ({a[:, 0].std():.4f})  d1 {a[:, 1].mean():.4f} ({a[:, 1].std():.4f})  "
          f"d2-3 {a[:, 2].mean():.4f} ({a[:, 2].std():.4f})  kernel {a[:, 3].mean():.3f} ms  launches {int(a[0, 4])}", flush=True)
