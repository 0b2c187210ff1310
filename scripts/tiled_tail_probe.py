"""DRB1-3123 x120: WHERE do the default sampler's larger errors at path distance 1 sit?  The 120 copies are identical, so every
adjacent step pair exists 120 times: for each pair of the fixture (by its place in the fixture) the squared relative error
averaged over the copies — systematic (the same junctions in every copy) or noise (some copies)?
    python scripts/tiled_tail_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import O, G, P, load, oracle_graph, oracle_params   # noqa: E402
from gfasort_amd import hip, quality as Q   # noqa: E402

copies = 120
g0 = load("DRB1-3123.gfa")
g = G.tile_series(g0, copies)
pos, _ = g.step_positions()
first = g.path_first_step.astype(np.int64)
sn = g.step_node.astype(np.int64)
N0 = g0.n_nodes
ctx = hip.Context(g)


def run(flags, seed=9399220):
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.seed = seed
    ctx.setup_1d(p, hip.make_config(flags=flags))
    ctx.init_positions()
    ctx.run()
    return ctx.download()


# adjacent pairs inside a copy, keyed by (path, rank within the copy)
f0 = g0.path_first_step.astype(np.int64)
keys, sa_all = [], []
for pth in range(g.n_paths):
    c0 = int(f0[pth + 1] - f0[pth])
    r = np.arange(c0 - 1)
    for c in range(copies):
        sa_all.append(first[pth] + c * c0 + r)
    keys.append(np.stack([np.full(c0 - 1, pth), r], axis=1))
keys = np.concatenate(keys)                                   # [n_pairs0, 2]
sa = np.concatenate(sa_all)
n0 = keys.shape[0]
d = (pos[sa + 1].astype(np.float64) - pos[sa].astype(np.float64))


def per_pair(x):
    ld = np.abs(x[sn[sa + 1]] - x[sn[sa]])
    e = (ld - d) / np.maximum(d, 1e-300)
    e[d == 0] = 0.0
    # layout: for path pth the copies are consecutive blocks of (c0 - 1) pairs
    out = np.zeros((n0, copies))
    o = 0
    k0 = 0
    for pth in range(g.n_paths):
        c0 = int(f0[pth + 1] - f0[pth]) - 1
        out[k0:k0 + c0, :] = e[o:o + c0 * copies].reshape(copies, c0).T
        o += c0 * copies
        k0 += c0
    return out


res = {}
for name, flags in (("reference streams", hip.F_BUNDLE(1)), ("default", 0), ("default seed+1000", 0)):
    x = run(flags, 9399220 + (1000 if "seed" in name else 0))
    res[name] = per_pair(x)
    E = res[name]
    ms = (E ** 2).mean(axis=1)                                 # per fixture pair, mean over copies
    tot = ms.sum()
    o = np.argsort(ms)[::-1]
    print(f"{name:20s} rms over all pairs {np.sqrt((E ** 2).mean()):.2f}; share of the mean square in the worst 10/100/1000 fixture pairs "
          f"(of {n0}): {ms[o[:10]].sum() / tot:.3f} {ms[o[:100]].sum() / tot:.3f} {ms[o[:1000]].sum() / tot:.3f}", flush=True)
    for k in o[:12]:
        pth, r = keys[k]
        s0 = f0[pth] + r
        print(f"    path {pth:2d} rank {r:4d}: nodes {g0.step_node[s0]:4d}->{g0.step_node[s0 + 1]:4d} len {g0.node_len[g0.step_node[s0]]:4d}; rel err over the copies: "
              f"mean {E[k].mean():8.1f} sd {E[k].std():7.1f} min {E[k].min():8.1f} max {E[k].max():8.1f}", flush=True)
a, b = res["reference streams"], res["default"]
ma, mb = (a ** 2).mean(axis=1), (b ** 2).mean(axis=1)
diff = mb - ma
o = np.argsort(diff)[::-1]
print(f"excess mean square of the default over reference streams: total {diff.sum() / n0:.1f}; in the 10/100/1000 pairs with the largest excess: "
      f"{diff[o[:10]].sum() / n0:.1f} {diff[o[:100]].sum() / n0:.1f} {diff[o[:1000]].sum() / n0:.1f}", flush=True)
for k in o[:15]:
    pth, r = keys[k]
    s0 = f0[pth] + r
    print(f"    path {pth:2d} rank {r:4d} nodes {g0.step_node[s0]:4d}->{g0.step_node[s0 + 1]:4d} len {g0.node_len[g0.step_node[s0]]:4d}: reference mean {a[k].mean():7.1f} sd {a[k].std():6.1f} | "
          f"default mean {b[k].mean():7.1f} sd {b[k].std():6.1f} | default other seed mean {res['default seed+1000'][k].mean():7.1f}", flush=True)
ctx.close()
