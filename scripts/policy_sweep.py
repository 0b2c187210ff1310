"""The auto policy's transition region, measured: bubble graphs of 16k...300k nodes (and two window graphs), `-p Y` defaults.
For every graph: reference streams (3 seeds -> mean profile) and the team kernel at B = 16 / 32 / 64 with runs of 1 / 16 / 64
trips (3 seeds each): worst ratio of the relative error over the octaves of path distance (distance 1 and 2-3 over ALL pairs),
ratio at distance 1, sampled-stress ratio, rate, and the number of INDEPENDENT leader draws per iteration
(updates / (B x trips of a run on this graph's paths x partners)).
    python scripts/policy_sweep.py [quick]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, quality as Q   # noqa: E402

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
graphs = [("bubbles", n, h) for n, h in ((12_500, 12), (25_000, 12), (50_000, 16), (100_000, 16), (230_000, 16))]
graphs += [("windows", 33_000, 16), ("windows", 131_000, 16)]
if quick:
    graphs = graphs[:2]
variants = [(16, 1), (32, 1), (64, 1), (64, 16), (64, 64)]
seeds = (9399220, 9400220, 9401220)


def run_trips(chain, cnt):
    room = cnt // 256
    if room < 2 or chain < 2:
        return 1
    return min(1 << (room.bit_length() - 1), chain)


print("graph | variant | streams | G upd/s | leaders per iteration | stress ratio | d1 ratio | worst octave ratio (octave) | inversions (windows)", flush=True)
for kind, n, h in graphs:
    g = G.synth_bubbles(n, h, 11) if kind == "bubbles" else G.synth_windows(n, h, n // 4, 11)
    p0 = P.YgsParams.from_graph(g, 0, 1).path_sgd
    cnt = int(g.path_step_counts().max())
    ctx = hip.Context(g)

    def one(flags, seed):
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.seed = seed
        ctx.setup_1d(p, hip.make_config(flags=flags))
        ctx.init_positions()
        ctx.run()
        st = ctx.stats()
        x = ctx.download()
        _, rms, _ = Q.stress_by_scale(g, x, 0, 600_000)
        inv = Q.inversions_vs_chain(g.node_ids[ctx.sort_order().astype(np.int64)].astype(np.int64)) if kind == "windows" else -1
        return rms, Q.sampled_stress(g, x, 0, 1_000_000), st, inv

    ref = [one(hip.F_BUNDLE(1), s) for s in seeds]
    pr = np.mean([r[0] for r in ref], axis=0)
    sr = float(np.mean([r[1] for r in ref]))
    spread = float(np.max(np.std([r[0] for r in ref], axis=0) / pr))
    st = ref[0][2]
    print(f"{kind} {g.n_nodes} nodes {g.n_paths} paths x {cnt} steps | reference streams | {st.n_streams} | "
          f"{st.term_updates / (st.kernel_ms * 1e-3) / 1e9:.1f} | - | 1 | 1 | (sd between seeds up to {spread:.3f}) | {ref[0][3]}", flush=True)
    auto = one(0, seeds[0])[2]
    print(f"    auto policy picks B = {auto.bundle}, K <= {auto.run_trips}, {auto.n_streams} streams", flush=True)
    for B, K in variants:
        res = [one(hip.F_BUNDLE(B) | hip.F_CHAIN(K), s) for s in seeds]
        pn = np.mean([r[0] for r in res], axis=0)
        ratio = pn / pr
        st = res[0][2]
        k_eff = run_trips(K, cnt) if B == 64 else 1
        partners = 2 if B == 64 else 1
        leaders = p0.min_term_updates / (B * k_eff * partners)
        print(f"    B {B:2d} K {K:2d} (runs of {k_eff:2d} trips) | {st.n_streams} | {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:.1f} | {leaders:.0f} | "
              f"{np.mean([r[1] for r in res]) / sr:.3f} | {ratio[0]:.3f} | {ratio.max():.3f} ({int(ratio.argmax())}) | "
              f"{max(r[3] for r in res)}", flush=True)
    ctx.close()
