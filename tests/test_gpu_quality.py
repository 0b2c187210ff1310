"""P2 on the graphs where it matters (SURVEY.md §7 hard part 1, §8c): large pangenome-like bubble graphs, the
product with ALL-DEFAULT flags against (i) the CPU oracle — the reference's own execution model: Hogwild worker threads
plus the 1 ms checker thread (oracle mode (a), sgd.rs:366-593) — and (ii) reference streams on the GPU (GFS_F_BUNDLE(1):
every lane is one reference worker thread, bit-exact against the oracle in test_gpu_parity.py).

Instruments (gfasort_amd/quality.py):
  * sampled stress, the reference's own formula (sgd.rs:1196-1283), 2M pairs (at 10k-200k pairs the value swings by
    +-10 % between two seeds of the SAME sampler on these graphs: a handful of short-range pairs dominates it);
  * the same relative error resolved by path distance (octaves of step distance) — stable to 1 % between seeds, and it
    shows WHERE two samplers differ;
  * measure_layout_quality's RMSE / MAE of the resulting sort (measure_layout_quality.rs:100-208, RNG-free);
  * Kendall tau and Spearman rho of the two rank orders after orienting (a 1D layout is mirror-invariant).
"""
import numpy as np
import pytest

from util import O, G, P, load, oracle_graph, oracle_params
from gfasort_amd import hip, quality as Q

pytestmark = pytest.mark.gpu


def _run_default(ctx, p, flags=0):
    ctx.setup_1d(p, hip.make_config(flags=flags))
    ctx.init_positions()
    ctx.run()
    return ctx.download(), ctx.stats()


def _profile(g, x):
    _, rms, cnt = Q.stress_by_scale(g, x, 0, 1_000_000)
    return rms


def _compare(g, og, x_ref, x_new, what, tol_stress=0.10, tol_octave=0.12):
    s_ref, s_new = O.stress_1d(og, x_ref, 2_000_000), O.stress_1d(og, x_new, 2_000_000)
    assert s_new <= (1.0 + tol_stress) * s_ref, (what, "sampled stress", s_ref, s_new)
    pr, pn = _profile(g, x_ref), _profile(g, x_new)
    worst = float(np.max(pn / pr))
    assert worst <= 1.0 + tol_octave, (what, "relative error by octave of path distance", " ".join(f"{v:.3f}" for v in pn / pr))
    o_ref, o_new = hip.sort_order(x_ref).astype(np.int64), hip.sort_order(x_new).astype(np.int64)
    q_ref, q_new = Q.layout_quality(g, o_ref), Q.layout_quality(g, o_new)
    assert q_new["rmse"] <= 1.05 * q_ref["rmse"] and q_new["mae"] <= 1.05 * q_ref["mae"], (what, q_ref, q_new)
    r_ref = Q.ranks_of(o_ref)
    r_new = Q.oriented(r_ref, Q.ranks_of(o_new))
    tau, rho = Q.kendall_tau(r_ref, r_new), Q.spearman_rho(r_ref, r_new)
    assert tau >= 0.99 and rho >= 0.99, (what, tau, rho)
    return dict(stress=(s_ref, s_new), worst_octave=worst, rmse=(q_ref["rmse"], q_new["rmse"]), tau=tau)


def test_default_flags_on_a_525k_node_bubble_graph_against_the_cpu_oracle_and_reference_streams():
    g = G.synth_bubbles(400_000, 24, 6)                         # 525 000 nodes, 24 haplotypes, 9.75e6 steps
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd                # the CLI's defaults: iter_max 100
    og = oracle_graph(g)
    ctx = hip.Context(g)
    x_def, st = _run_default(ctx, p)
    assert st.bundle == 64 and st.run_trips == 64 and st.launches == 1        # what the library picks by itself
    assert st.term_updates == (p.iter_max + 1) * p.min_term_updates
    x_b1, st1 = _run_default(ctx, p, hip.F_BUNDLE(1))
    assert st1.bundle == 1 and st1.term_updates == st.term_updates
    _compare(g, og, x_b1, x_def, "default flags vs GPU reference streams")
    # More streams than the chip holds at once (this graph's 33 KB zeta table in LDS = 4 workgroups of 256 per CU = 4096
    # waves; 4101 asked for): the barrier-free fused launch would let the 5 late waves walk the whole schedule after the
    # others are done (relative error 64 at path distance 1 when it did), so the library runs one launch per iteration.
    before = ctx.stats().launches                               # (counted over the context's life)
    ctx.setup_1d(p, hip.make_config(n_streams=262_464))
    ctx.init_positions()
    ctx.run()
    x_many, stm = ctx.download(), ctx.stats()
    ctx.close()
    assert stm.n_streams == 262_464 and stm.bundle == 64 and stm.launches - before == p.iter_max + 1 and stm.term_updates == st.term_updates
    _compare(g, og, x_b1, x_many, "more streams than resident workgroups vs GPU reference streams")
    # Work pools: the precision at path distance 1 does not depend on the stream count.  With round 1's launch (a fixed quota per
    # wave, free-running waves that drift apart in the schedule) 209 920 streams was a bad count on this graph.  Measured over
    # ALL 9.75e6 adjacent step pairs (Q.short_range_error): 0.1928-0.1937 with pools at either count over three seeds, 0.204
    # for reference streams (profiles/r03/d1_outliers.log).  (Round 2 measured a 50 000-pair sample of it and saw 0.189-0.218:
    # the squared relative error at distance 1 is heavy-tailed — the worst ten pairs of 9.75e6 carry 0.6-2 % of the mean
    # square — and a sample swings by +-7 % with whether it hits one of them; the same seed gave the same "outlier" twice
    # because the same pairs were sampled: profiles/r03/pool_seed_study.log.)
    ctx = hip.Context(g)
    d1 = {}
    for name, fl in (("pools", 0), ("free-running", hip.F_DBG_FREE_RUNNING)):
        ctx.setup_1d(p, hip.make_config(n_streams=209_920, flags=fl))
        ctx.init_positions()
        ctx.run()
        assert ctx.stats().term_updates == st.term_updates
        d1[name] = Q.short_range_error(g, ctx.download(), 0, (1,))["rms"]
    ctx.close()
    d1_ref, d1_def = Q.short_range_error(g, x_b1, 0, (1,))["rms"], Q.short_range_error(g, x_def, 0, (1,))["rms"]
    assert d1["pools"] <= 1.02 * d1_ref and d1_def <= 1.02 * d1_ref, (d1, d1_def, d1_ref)
    assert abs(d1["pools"] - d1_def) <= 0.015 * d1_def, (d1, d1_def)            # the stream count does not matter
    assert d1["free-running"] >= 1.08 * d1["pools"], d1                          # ... and the pools are what makes it so
    # the CPU oracle, executed as the reference executes: worker threads + checker thread (flat arrays, all host cores)
    import os
    op = oracle_params(p)
    op.nthreads = max(2, min(16, len(os.sched_getaffinity(0))))
    x_cpu = O.init_positions(og)
    rc, cst = O.sgd_1d_threads(og, op, x_cpu, flat=1)
    assert rc == 0 and cst.iterations >= p.iter_max
    # the checker thread overshoots every iteration by what the workers do in 1 ms: equal update counts to a few %
    assert 0.99 <= cst.term_updates / st.term_updates < 1.15, cst.term_updates / st.term_updates
    _compare(g, og, x_cpu, x_def, "default flags vs the CPU oracle (threads)")
    _compare(g, og, x_cpu, x_b1, "GPU reference streams vs the CPU oracle (threads)")


def test_default_flags_on_a_2m_node_bubble_graph_against_reference_streams():
    g = G.synth_bubbles(1_500_000, 32, 7)                       # 1 968 750 nodes, 32 haplotypes, 4.9e7 steps
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    og = oracle_graph(g)
    ctx = hip.Context(g)
    x_def, st = _run_default(ctx, p)
    assert st.bundle == 64 and st.run_trips == 64 and st.term_updates == (p.iter_max + 1) * p.min_term_updates
    x_b1, st1 = _run_default(ctx, p, hip.F_BUNDLE(1))
    ctx.close()
    res = _compare(g, og, x_b1, x_def, "default flags vs GPU reference streams, 2M nodes")
    # the round-1 sampler (short runs, first colour only) is what these thresholds exist to catch: +40-55 % stress
    ctx = hip.Context(g)
    x_r1, _ = _run_default(ctx, p, hip.F_BUNDLE(64) | hip.F_CHAIN(1) | hip.F_DBG_ONE_COLOUR | hip.F_ONE_PARTNER | hip.F_DBG_FREE_RUNNING)
    ctx.close()
    with pytest.raises(AssertionError):
        _compare(g, og, x_b1, x_r1, "round-1 sampler")
    assert res["tau"] > 0.9999


# ---- the DEFAULT layout kernel (sgdnd_team_kernel<D,64>: one set of end flips per run where sgd.rs:1062,1071 draw them per
# ---- term, two partners per leader, twin trips) on large graphs, against the CPU oracle and GPU reference streams -----------
def _layout_profile(g, c, dims):
    _, rms, cnt = Q.stress_by_scale(g, c, dims, 1_000_000)
    return rms


def _end_to_end(g, c, dims):
    """|distance between a node's two ends - its length| (what a layout of a sequence graph must get right first)"""
    cc = np.asarray(c).reshape(-1, 2, dims)
    d = np.sqrt(((cc[:, 0, :] - cc[:, 1, :]) ** 2).sum(axis=1))
    err = np.abs(d - g.node_len)
    return float(np.median(err)), float(np.mean(err))


def _compare_layout(g, og, dims, c_ref, c_new, what, tol_stress=0.10, tol_octave=0.12, tol_e2e=0.10):
    s_ref, s_new = O.layout_stress(og, dims, c_ref, 2_000_000), O.layout_stress(og, dims, c_new, 2_000_000)
    assert s_new <= (1.0 + tol_stress) * s_ref, (what, "sampled layout stress", s_ref, s_new)
    pr, pn = _layout_profile(g, c_ref, dims), _layout_profile(g, c_new, dims)
    worst = float(np.max(pn / pr))
    assert worst <= 1.0 + tol_octave, (what, "relative error by octave of path distance", " ".join(f"{v:.3f}" for v in pn / pr))
    (m_ref, a_ref), (m_new, a_new) = _end_to_end(g, c_ref, dims), _end_to_end(g, c_new, dims)
    assert m_new <= (1.0 + tol_e2e) * m_ref + 0.02 and a_new <= (1.0 + tol_e2e) * a_ref + 0.02, (what, "node end-to-end distance vs length", (m_ref, a_ref), (m_new, a_new))
    return dict(stress=(s_ref, s_new), worst_octave=worst, end_to_end=((m_ref, a_ref), (m_new, a_new)))


def test_default_layout_flags_on_a_525k_node_bubble_graph_against_the_cpu_oracle_and_reference_streams():
    """`-p L --dimensions 2` with ALL-DEFAULT flags on the 525k-node bubble graph: the team kernel the library picks
    (B = 64, runs of 16 trips, two partners) against (i) the CPU oracle run as the reference runs its layout — worker
    threads + checker thread, sgd.rs:925-1164 — and (ii) reference streams on the GPU, at equal update counts from the same
    start: sampled layout stress (2M pairs), relative error per octave of path distance, node end-to-end distance."""
    import os
    from gfasort_amd import sgd as S
    g = G.synth_bubbles(400_000, 24, 6)
    p = P.LayoutSGDParams.from_graph(g, 2, 1)                    # the CLI's defaults: 31 iterations of 10 x steps
    og = oracle_graph(g)
    c0 = S.default_layout_init(g, 2, p.seed)
    rc, c_def, st = hip.path_linear_sgd_layout_raw(g, p, c0)
    assert rc == 0 and st.bundle == 64 and st.term_updates == (p.iter_max + 1) * p.min_term_updates
    rc, c_b1, st1 = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=hip.F_BUNDLE(1)))
    assert rc == 0 and st1.bundle == 1 and st1.term_updates == st.term_updates
    res = _compare_layout(g, og, 2, c_b1, c_def, "default layout flags vs GPU reference streams")
    op = oracle_params(p)
    op.nthreads = max(2, min(16, len(os.sched_getaffinity(0))))
    c_cpu = c0.copy()
    rc, cst = O.sgd_nd_threads(og, op, c_cpu, flat=1)
    assert rc == 0 and cst.iterations >= p.iter_max
    assert 0.99 <= cst.term_updates / st.term_updates < 1.15, cst.term_updates / st.term_updates
    _compare_layout(g, og, 2, c_cpu, c_def, "default layout flags vs the CPU oracle (threads)")
    _compare_layout(g, og, 2, c_cpu, c_b1, "GPU layout reference streams vs the CPU oracle (threads)")
    s0 = O.layout_stress(og, 2, c0, 200_000)
    assert res["stress"][1] < 0.1 * s0


def test_default_layout_flags_in_3_dimensions_against_reference_streams():
    from gfasort_amd import sgd as S
    g = G.synth_bubbles(150_000, 16, 9)                          # 196 875 nodes, 16 haplotypes
    p = P.LayoutSGDParams.from_graph(g, 3, 1)
    og = oracle_graph(g)
    c0 = S.default_layout_init(g, 3, p.seed)
    rc, c_def, st = hip.path_linear_sgd_layout_raw(g, p, c0)
    assert rc == 0 and st.bundle == 64 and st.term_updates == (p.iter_max + 1) * p.min_term_updates
    rc, c_b1, st1 = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=hip.F_BUNDLE(1)))
    assert rc == 0 and st1.bundle == 1
    _compare_layout(g, og, 3, c_b1, c_def, "default 3-D layout flags vs GPU reference streams")


# ---- default flags on REAL pangenome structure at scale: the reference's DRB1-3123 fixture tiled in series -----------------
@pytest.mark.parametrize("shuffle_seed", [None, 17])
def test_default_flags_on_drb1_tiled_in_series(shuffle_seed):
    """tests/data/DRB1-3123.gfa (the fixture of integration_tests.rs:147-172: 4 955 nodes, nested bubbles, kilobase insertions,
    one haplotype on the reverse strand) 120 times in series: 594 600 nodes, 12 haplotype paths of up to 372 000 steps,
    371 520 reverse steps — from the fixture's own node order and from block-shuffled S lines (unsorted input is what gfasort
    exists for).  The auto policy must pick the team kernel (B = 64, long runs).

    What was found (profiles/r03/tiled_probe*.log).  On this graph the reference's default schedule (--iter-max 100) does NOT
    converge, for any sampler: reference streams and the CPU oracle end at a relative error of 26-29 (2 800 %) at path
    distance 1 and a rank RMSE of 195 bp; a schedule of 300 iterations ends at 12 and 125 bp, one of 1000 at 7.4 and 104 bp.
    (a) At the schedule lengths where the layout converges the default sampler is at parity or BETTER in every octave and in
        RMSE — asserted here at --iter-max 300 with the thresholds of the bubble-graph tests, against reference streams and
        against the CPU oracle run as the reference runs (threads + checker).
    (b) At the unconverged point of --iter-max 100 it is BEHIND: +60-70 % at distance 1, +10-16 % from 32 steps up, RMSE +22 %,
        sampled stress +4-12 % — whatever the run length (K = 1...128), the partners, the stream count (16k...262k) or the launch:
        correlated terms are fewer independent samples per update, and an annealing schedule that is too short for the graph
        shows it.  Asserted as measured (bounds with ~15 % of head room), so that a regression is caught and the gap stays on
        record; `GFS_F_BUNDLE(1)` / `gfasort_hip --reference-sampler` runs the reference's own sampler (8x slower)."""
    import os
    g = G.tile_series(load("DRB1-3123.gfa"), 120, shuffle_seed=shuffle_seed)
    assert g.n_nodes == 594_600 and int(g.step_is_rev.sum()) == 371_520
    og = oracle_graph(g)
    ctx = hip.Context(g)
    res = {}
    for iter_max in (300, 100):
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = iter_max
        x_def, st = _run_default(ctx, p)
        assert st.bundle == 64 and st.run_trips == 64 and st.term_updates == (p.iter_max + 1) * p.min_term_updates
        l0 = st.launches
        x_b1, st1 = _run_default(ctx, p, hip.F_BUNDLE(1))
        assert st1.bundle == 1 and st1.launches - l0 == 1 and st1.term_updates == st.term_updates     # (launches: over the context's life)
        res[iter_max] = (x_def, x_b1)
    ctx.close()
    # (a) converged schedule: parity
    x_def, x_b1 = res[300]
    _compare(g, og, x_b1, x_def, "DRB1 x120 --iter-max 300, default flags vs GPU reference streams")
    if shuffle_seed is None:
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = 300
        op = oracle_params(p)
        op.nthreads = max(2, min(16, len(os.sched_getaffinity(0))))
        x_cpu = O.init_positions(og)
        rc, cst = O.sgd_1d_threads(og, op, x_cpu, flat=1)
        assert rc == 0 and cst.iterations >= p.iter_max
        assert 0.99 <= cst.term_updates / (301 * p.min_term_updates) < 1.25
        _compare(g, og, x_cpu, x_def, "DRB1 x120 --iter-max 300, default flags vs the CPU oracle (threads)")
    # (b) the reference's default schedule: behind, by a bounded amount
    x_def, x_b1 = res[100]
    s_ref, s_def = O.stress_1d(og, x_b1, 2_000_000), O.stress_1d(og, x_def, 2_000_000)
    assert s_def <= 1.20 * s_ref, (s_ref, s_def)               # measured +4...12 % (the sampled stress itself: +-3 % per run)
    pr, pn = _profile(g, x_b1), _profile(g, x_def)
    ratio = pn / pr
    assert ratio[0] <= 2.0 and np.max(ratio[1:]) <= 1.30, np.round(ratio, 3).tolist()
    q_ref = Q.layout_quality(g, hip.sort_order(x_b1).astype(np.int64))
    q_def = Q.layout_quality(g, hip.sort_order(x_def).astype(np.int64))
    assert q_def["rmse"] <= 1.40 * q_ref["rmse"], (q_ref, q_def)
    r_ref = Q.ranks_of(hip.sort_order(x_b1).astype(np.int64))
    r_def = Q.oriented(r_ref, Q.ranks_of(hip.sort_order(x_def).astype(np.int64)))
    assert Q.kendall_tau(r_ref, r_def) >= 0.99


def test_default_layout_flags_on_drb1_tiled_in_series():
    """`-p L --dimensions 2` on DRB1-3123 x120 (see the test above): the default layout kernel against reference streams from the
    same starts.  ONE run of either sampler is not a yardstick on this graph: at --layout-iter 90 a run ends in one of two states,
    ~9 % apart in stress and in every octave of path distance (reference streams: 0.2255, 0.2257, 0.2256, 0.2096 over four seeds;
    the default kernel 0.2109, 0.2267, 0.2238, 0.2179 — profiles/r03/tiled_layout_seed_study.log; round 3's first form of this
    test compared single runs and passed or failed by which state each had drawn).  So: four seeds each, the MEANS compared —
    measured 0.92...1.045 per octave at 90 iterations.  As for the sort, the reference's default schedule (--layout-iter 30) does
    not converge on this graph (relative error 29 at path distance 1 for reference streams) and the run sampler is behind there
    — milder than in 1D: +12 % at distance 1, +9...13 % from 32 steps up (profiles/r03/layout_partner_probe.log)."""
    from gfasort_amd import sgd as S
    g = G.tile_series(load("DRB1-3123.gfa"), 120)
    og = oracle_graph(g)
    for iters in (90, 30):
        stress, prof, e2e = {"def": [], "ref": []}, {"def": [], "ref": []}, {"def": [], "ref": []}
        for seed in range(4):
            p = P.LayoutSGDParams.from_graph(g, 2, 1)
            p.iter_max = iters
            p.seed = p.seed + 1000 * seed
            c0 = S.default_layout_init(g, 2, p.seed)
            rc, c_def, st = hip.path_linear_sgd_layout_raw(g, p, c0)
            assert rc == 0 and st.bundle == 64 and st.term_updates == (p.iter_max + 1) * p.min_term_updates
            rc, c_b1, st1 = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=hip.F_BUNDLE(1)))
            assert rc == 0 and st1.bundle == 1 and st1.term_updates == st.term_updates
            for k, c in (("def", c_def), ("ref", c_b1)):
                stress[k].append(O.layout_stress(og, 2, c, 2_000_000))
                prof[k].append(_layout_profile(g, c, 2))
                e2e[k].append(_end_to_end(g, c, 2))
        s_def, s_ref = float(np.mean(stress["def"])), float(np.mean(stress["ref"]))
        ratio = np.mean(prof["def"], axis=0) / np.mean(prof["ref"], axis=0)
        (m_def, a_def), (m_ref, a_ref) = np.mean(e2e["def"], axis=0), np.mean(e2e["ref"], axis=0)
        what = (iters, stress, " ".join(f"{v:.3f}" for v in ratio), (m_ref, a_ref), (m_def, a_def))
        if iters == 90:
            # (stress and every octave at parity; the MEDIAN of |end-to-end distance - node length| is still ~0.1 bp behind —
            # 1.01-1.16 against 0.90-1.01 bp, the mean 13.9 against 14.1 — the short-range side of the same lag)
            # (bounds: four runs of one sampler all in the worse state against four of the other all in the better one are 9 % apart
            # before any difference between the samplers: 1.09 x the measured 1.045)
            assert s_def <= 1.12 * s_ref and ratio.max() <= 1.18, what
            assert m_def <= 1.30 * m_ref + 0.02 and a_def <= 1.30 * a_ref + 0.02, what
        else:
            assert s_def <= 1.12 * s_ref and ratio.max() <= 1.25, what


def test_more_than_4_million_paths():
    """Round 1 kept 22 bits for the path id in a step record (n_paths < 2^22); the crowding exponents now live in the
    spare top bits of the position's high word.  4.3M two-step paths over a chain: exact counts, and the chain sorts."""
    n, n_paths = 1_000_000, 4_300_000
    rng = np.random.default_rng(5)
    start = rng.integers(0, n - 1, size=n_paths).astype(np.uint32)
    steps = np.stack([start, start + 1], axis=1).reshape(-1)
    order = np.arange(n)
    for b in range(0, n, 64):                                   # block-shuffled S lines like the other generators
        rng.shuffle(order[b:b + 64])
    inv = np.empty(n, dtype=np.int64)
    inv[order] = np.arange(n)
    g = G.FlatGraph(node_len=np.full(n, 3, dtype=np.uint32), step_node=inv[steps].astype(np.uint32),
                    step_is_rev=np.zeros(steps.shape[0], np.uint8),
                    path_first_step=(np.arange(n_paths + 1, dtype=np.uint64) * 2), node_ids=(order + 1).astype(np.uint64),
                    path_names=[])
    p = P.PathSGDParams()
    p.min_term_updates, p.eta_max, p.space, p.iter_max = g.n_steps, 4.0, 6, 30
    # every sampled term must pair the two steps of ONE path: neighbours on the chain, 3 bp apart.  A path id cut to 22
    # bits would look up another path's record and pair a step with a stranger.
    ctx = hip.Context(g)
    ctx.setup_1d(p, hip.make_config(trace_per_stream=6))
    ctx.init_positions()
    ctx.run_iteration(0)
    tr, counts = ctx.trace()
    ctx.close()
    took = tr[np.arange(tr.shape[1])[None, :] < counts[:, None]]
    assert took.shape[0] > 1_000_000
    place = order                                                # dense index k holds chain node order[k]
    assert np.all(np.abs(place[took["i"].astype(np.int64)] - place[took["j"].astype(np.int64)]) == 1)
    assert np.all(took["d_ij"] == 3.0)
    rc, x, st = hip.path_linear_sgd_raw(g, p)
    assert rc == 0 and st.term_updates == 31 * g.n_steps and np.isfinite(x).all() and st.bundle == 1
    # the covered adjacent pairs are pulled towards 3 bp apart (only neighbour terms exist, so the chain relaxes slowly)
    xs = np.empty(n)
    xs[order] = x                                               # by chain position
    covered = np.zeros(n - 1, dtype=bool)
    covered[np.unique(start)] = True
    gap = np.abs(np.diff(xs))[covered]
    x0 = hip.init_positions(g)
    xs0 = np.empty(n)
    xs0[order] = x0
    gap0 = np.abs(np.diff(xs0))[covered]
    assert np.median(np.abs(gap - 3.0)) < 0.25 * np.median(np.abs(gap0 - 3.0)), (np.median(np.abs(gap - 3.0)), np.median(np.abs(gap0 - 3.0)))


def test_a_context_with_nothing_to_do_keeps_a_position_replica():
    """No path of more than one step: the reference returns before any update (sgd.rs:250-261) and the one-shot call
    says GFS_NOTHING_TO_DO; a resident context still owns a full-length replica (a multi-GPU rank whose shard has no
    multi-step path uploads, merges and downloads like its peers)."""
    n = 1000
    g = G.FlatGraph(node_len=np.full(n, 2, dtype=np.uint32), step_node=np.arange(10, dtype=np.uint32),
                    step_is_rev=np.zeros(10, np.uint8), path_first_step=np.arange(11, dtype=np.uint64),
                    node_ids=np.arange(1, n + 1, dtype=np.uint64), path_names=[f"p{k}" for k in range(10)])
    p = P.PathSGDParams()
    ctx = hip.Context(g)
    assert ctx.setup_1d(p, hip.make_config()) == hip.NOTHING_TO_DO
    assert ctx.positions_len() == n
    x0 = np.linspace(0.0, 1.0, n)
    ctx.upload(x0)
    assert ctx.run() == hip.NOTHING_TO_DO and ctx.run_iteration(0) == hip.NOTHING_TO_DO
    assert np.array_equal(ctx.download(), x0)
    ctx.init_positions()
    assert np.array_equal(ctx.download(), hip.init_positions(g))
    assert ctx.stats().term_updates == 0
    ctx.close()
    rc, x, st = hip.path_linear_sgd_raw(g, p, x=x0.copy())
    assert rc == hip.NOTHING_TO_DO and np.array_equal(x, x0)
