"""Where do two samplers differ?  Large bubble graphs, default -p Y run, per sampler variant and seed:
sampled stress (oracle's restatement of sgd.rs:1196, 200k pairs), relative error per octave of step distance,
measure_layout_quality RMSE of the resulting sort, Kendall tau against the first reference-stream run.

usage: quality_probe.py [small|large|both] [variant ...]   variants: name=flags e.g. B1, B8, B64, or any key of VARIANTS
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip, quality as Q
from oracle import oracle as O

VARIANTS = {
    "B1": hip.F_BUNDLE(1),
    "B8": hip.F_BUNDLE(8),
    "B16": hip.F_BUNDLE(16),
    "B32": hip.F_BUNDLE(32),
    "B64": hip.F_BUNDLE(64),
    "auto": 0,
}
VARIANTS["auto_p1"] = hip.F_ONE_PARTNER                     # one partner draw per leader (the sampler before twin trips)
VARIANTS["auto_notwin"] = hip.F_DBG_NO_TWIN_TRIP            # two partners, as two trips each
VARIANTS["auto_nofuse"] = hip.F_NO_FUSE                      # one launch per iteration
VARIANTS["auto_p1_nofuse"] = hip.F_NO_FUSE | hip.F_ONE_PARTNER
VARIANTS["auto_free"] = hip.F_DBG_FREE_RUNNING               # fixed quota per wave, free-running waves (round 1's fused launch)
VARIANTS["auto_p1_free"] = hip.F_DBG_FREE_RUNNING | hip.F_ONE_PARTNER
VARIANTS["B64_r1"] = hip.F_BUNDLE(64) | hip.F_CHAIN(1) | hip.F_DBG_ONE_COLOUR | hip.F_ONE_PARTNER | hip.F_DBG_FREE_RUNNING      # the round-1 sampler
for _k in (1, 2, 4, 8, 16, 32, 64):
    VARIANTS[f"B64_k{_k}"] = hip.F_BUNDLE(64) | hip.F_CHAIN(_k)
    VARIANTS[f"B64_k{_k}_nofusedtrip"] = hip.F_BUNDLE(64) | hip.F_CHAIN(_k) | hip.F_DBG_NO_FUSED_TRIP
    VARIANTS[f"B64_k{_k}_noalign"] = hip.F_BUNDLE(64) | hip.F_CHAIN(_k) | hip.F_DBG_NO_ALIGN


def run(ctx, p, flags, T=0):
    ctx.setup_1d(p, hip.make_config(n_streams=T, flags=flags))
    ctx.init_positions()
    ctx.run()
    st = ctx.stats()
    return ctx.download(), st


def study(name, g, variants, seeds=2, chain=False, iter_max=None):
    og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    if iter_max:
        p.iter_max = iter_max
    print(f"== {name}: nodes {g.n_nodes} steps {g.n_steps} paths {g.n_paths} iter_max {p.iter_max}", flush=True)
    ctx = hip.Context(g)
    ref_rank = None
    for v in variants:
        for sd in range(seeds):
            p.seed = 9399220 + 1000 * sd
            t0 = time.time()
            name, _, T = v.partition("@")                     # "variant@streams"
            x, st = run(ctx, p, VARIANTS[name], int(T) if T else 0)
            rate = st.term_updates / (st.kernel_ms * 1e-3) / 1e9
            s = O.stress_1d(og, x, 200000)
            s2m = O.stress_1d(og, x, 2_000_000)
            _, rms, cnt = Q.stress_by_scale(g, x, 0, 1_000_000)
            order = hip.sort_order(x).astype(np.int64)
            lq = Q.layout_quality(g, order)
            rank = Q.ranks_of(order)
            if ref_rank is None:
                ref_rank = rank
                tau = 1.0
            else:
                tau = Q.kendall_tau(ref_rank, Q.oriented(ref_rank, rank))
            if chain:
                print(f"  {v:22s} seed {sd} bundle {st.bundle:2d} k {st.run_trips:2d} streams {st.n_streams:6d} {rate:6.2f} G/s  "
                      f"inversions vs chain {Q.inversions_vs_chain(g.node_ids[order].astype(np.int64))}  stress {s:.3e}", flush=True)
                continue
            print(f"  {v:10s} seed {sd} bundle {st.bundle:2d} streams {st.n_streams:6d} launches {st.launches} {rate:6.2f} G/s  stress {s:.6f} (2M pairs: {s2m:.6f})  "
                  f"rmse {lq['rmse']:.3f} mae {lq['mae']:.3f}  tau_vs_first {tau:.6f}  ({time.time() - t0:.0f} s)", flush=True)
            print("      by octave: " + " ".join(f"{r:.4f}" for r in rms), flush=True)
    ctx.close()


def study_layout(name, g, variants, dims=2, seeds=2):
    """The same for path_linear_sgd_layout: relative error of the '+' end distances per octave."""
    from gfasort_amd import sgd as S
    og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
    p = P.LayoutSGDParams.from_graph(g, dims, 1)
    print(f"== layout D={dims} {name}: nodes {g.n_nodes} steps {g.n_steps} iter_max {p.iter_max} updates/iter {p.min_term_updates}", flush=True)
    c0 = S.default_layout_init(g, dims, 9399220)
    for v in variants:
        for sd in range(seeds):
            p.seed = 9399220 + 1000 * sd
            name_v, _, T = v.partition("@")
            rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(n_streams=int(T) if T else 0, flags=VARIANTS[name_v]))
            if os.environ.get("LAYOUT_RATE_ONLY"):
                print(f"  {v:10s} seed {sd} k {st.run_trips} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:6.2f} G/s", flush=True)
                continue
            rate = st.term_updates / (st.kernel_ms * 1e-3) / 1e9
            s = O.layout_stress(og, dims, c, 200000)
            _, rms, cnt = Q.stress_by_scale(g, c, dims, 1_000_000)
            print(f"  {v:10s} seed {sd} bundle {st.bundle:2d} streams {st.n_streams:6d} {rate:6.2f} G/s  stress {s:.6f}", flush=True)
            print("      by octave: " + " ".join(f"{r:.4f}" for r in rms), flush=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    variants = sys.argv[2:] or ["B1", "B8", "B64"]
    if which in ("small", "both"):
        study("bubbles 400k sites x 24 hap", G.synth_bubbles(400_000, 24, 6), variants)
    if which == "medium":
        for sites, haps, seed in ((20_000, 16, 5), (60_000, 16, 8), (150_000, 24, 9)):
            study(f"bubbles {sites} sites x {haps} hap", G.synth_bubbles(sites, haps, seed), variants)
    if which == "layout":
        study_layout("bubbles 400k sites x 24 hap", G.synth_bubbles(400_000, 24, 6), variants)
    if which == "c3":
        study("C3 windows(1M, 64, 156250, 2) --iter-max 200", G.synth_windows(1_000_000, 64, 156_250, 2), variants, chain=True, iter_max=200)
    if which in ("large", "both"):
        study("bubbles 1.5M sites x 32 hap", G.synth_bubbles(1_500_000, 32, 7), variants)


if __name__ == "__main__":
    main()
