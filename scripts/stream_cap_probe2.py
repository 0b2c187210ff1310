"""Medium graphs (66k...300k nodes): the team kernel's streams-per-node bound (capi.hip auto_stream_count: three streams per 4
nodes) leaves the chip partly empty there.  Default flags at 0.75 / 1.0 / 1.25 / 1.5 / 2.0 streams per node, two seeds each:
rate, and the relative error per octave of path distance against reference streams (distance 1 and 2-3 over all pairs).
    python scripts/stream_cap_probe2.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gfasort_amd import graph as G, params as P, hip, quality as Q   # noqa: E402

seeds = (9399220, 9400220)
print("graph | streams per node | streams | G upd/s | stress ratio | d1 ratio | 2-3 ratio | worst octave ratio (octave)", flush=True)
for n, h in ((50_000, 16), (100_000, 16), (230_000, 16)):
    g = G.synth_bubbles(n, h, 11)
    ctx = hip.Context(g)

    def one(flags, T, seed):
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.seed = seed
        ctx.setup_1d(p, hip.make_config(n_streams=T, flags=flags))
        ctx.init_positions()
        ctx.run()
        st = ctx.stats()
        x = ctx.download()
        return Q.stress_by_scale(g, x, 0, 600_000)[1], Q.sampled_stress(g, x, 0, 1_000_000), st

    ref = [one(hip.F_BUNDLE(1), 0, s) for s in seeds + (9401220,)]
    pr, sr = np.mean([r[0] for r in ref], axis=0), float(np.mean([r[1] for r in ref]))
    print(f"bubbles {g.n_nodes} nodes: reference streams {ref[0][2].n_streams} streams, {ref[0][2].term_updates / (ref[0][2].kernel_ms * 1e-3) / 1e9:.1f} G upd/s", flush=True)
    for per_node in (0.75, 1.0, 1.25, 1.5, 2.0):
        T = int(g.n_nodes * per_node) // 256 * 256
        res = [one(0, T, s) for s in seeds]
        ratio = np.mean([r[0] for r in res], axis=0) / pr
        st = res[0][2]
        print(f"    {per_node:4.2f} | {st.n_streams:7d} | {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.1f} | {np.mean([r[1] for r in res]) / sr:.3f} | "
              f"{ratio[0]:.3f} | {ratio[1]:.3f} | {ratio.max():.3f} ({int(ratio.argmax())}) | launches {st.launches}", flush=True)
    ctx.close()
