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

from util import O, G, P, oracle_graph, oracle_params
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
    assert worst <= 1.0 + tol_octave, (what, "relative error by octave of path distance", np.round(pn / pr, 3).tolist())
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
    # wave, free-running waves that drift apart in the schedule) 209 920 streams was a bad count on this graph: relative error
    # 0.217-0.228 at distance 1 against reference streams' 0.194; with the pools 0.19 at every count (profiles/r02/pacing.log).
    ctx = hip.Context(g)
    prof = {}
    for name, fl in (("pools", 0), ("free-running", hip.F_DBG_FREE_RUNNING)):
        ctx.setup_1d(p, hip.make_config(n_streams=209_920, flags=fl))
        ctx.init_positions()
        ctx.run()
        assert ctx.stats().term_updates == st.term_updates
        prof[name] = _profile(g, ctx.download())
    ctx.close()
    d1_ref = _profile(g, x_b1)[0]
    # pools: 0.189-0.205 over many runs (0.191 typical); free-running: 0.213-0.228.  Only the first is asserted — with a margin
    # for the run-to-run spread of a concurrent kernel; the second is in the message for whoever reads a failure.
    assert prof["pools"][0] <= 1.10 * d1_ref, (prof["pools"][0], d1_ref, "free-running:", prof["free-running"][0])
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
