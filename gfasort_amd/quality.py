"""Quality instruments for 1D sorts and nD layouts — host-side numpy restatements of the reference's
diagnostic binaries, RNG-free wherever the reference is (SURVEY.md §8c(4), §7 hard part 1):

* `layout_quality`   — `src/bin/measure_layout_quality.rs:100-208`: consecutive path steps, 1D distance of the
                        SORTED graph (prefix sum of node lengths in rank order) against the length of the first node.
* `sampled_stress`   — the formula of `calculate_layout_stress` (`src/sgd.rs:1196-1283`) / `compare_layouts.rs:156-255`
                        on numpy's generator (the reference's own sample stream, seed 12345 on rand 0.9, is restated
                        in oracle/ only; this one is for the product side: bench.py and the CLI-level checks).
* `stress_by_scale`  — the same relative error, but with the step distance drawn log-uniformly and reported per
                        octave, so that one can see WHERE (which path distances) two layouts differ; the plain sampled
                        stress draws uniform pairs and is dominated by the handful of short-range pairs it happens to hit.
* `kendall_tau`, `spearman_rho`, `oriented` — rank agreement of two sorts (a 1D layout is mirror-invariant).
* `inversions_vs_chain` — exact count of adjacent inversions against a known chain order (P1 graphs).
"""
import numpy as np

from .graph import NO_NODE


# ---- measure_layout_quality.rs -----------------------------------------------------------------
def layout_quality(g, order):
    """order[r] = dense index of the node of rank r (what `gfs_path_sgd_sort` returns and `apply_ordering` turns into
    ids 1..N).  Returns dict(mse, rmse, mae, relative_error, steps) exactly as measure_layout_quality.rs:100-208
    computes them on the sorted GFA: node position = prefix sum of sequence lengths in id order (:100-108); for every
    consecutive pair of path steps genomic distance = len(node_a) (:139-144), layout distance = |pos_b - pos_a| (:149-151)."""
    order = np.asarray(order, dtype=np.int64)
    n = g.n_nodes
    node_len = g.node_len.astype(np.float64)
    pos = np.zeros(n, dtype=np.float64)
    pos[order] = np.concatenate([[0.0], np.cumsum(node_len[order])[:-1]])
    first = g.path_first_step.astype(np.int64)
    sn = g.step_node.astype(np.int64)
    S = sn.shape[0]
    if S < 2:
        return dict(mse=0.0, rmse=0.0, mae=0.0, relative_error=0.0, steps=0)
    is_last = np.zeros(S, dtype=bool)
    is_last[first[1:][first[1:] > first[:-1]] - 1] = True
    a = np.nonzero(~is_last)[0]
    na, nb = sn[a], sn[a + 1]
    ok = na != NO_NODE                                   # :139-144 `continue` when node A is absent
    na, nb = na[ok], nb[ok]
    gd = node_len[na]
    pb = np.where(nb == NO_NODE, 0.0, pos[np.minimum(nb, n - 1)])     # :149-150 unwrap_or(0.0)
    err = np.abs(pb - pos[na]) - gd
    steps = int(err.shape[0])
    if steps == 0:
        return dict(mse=0.0, rmse=0.0, mae=0.0, relative_error=0.0, steps=0)
    mse = float(np.mean(err * err))
    mae = float(np.mean(np.abs(err)))
    return dict(mse=mse, rmse=float(np.sqrt(mse)), mae=mae,
                relative_error=float(mae / (float(gd.sum()) / steps)) if gd.sum() > 0 else 0.0, steps=steps)


# ---- calculate_layout_stress (formula), numpy sample stream --------------------------------------
def _pair_errors(g, coords, dims, sa, sb, pos):
    sn = g.step_node.astype(np.int64)
    d = np.abs(pos[sa].astype(np.float64) - pos[sb].astype(np.float64))
    ia, ib = sn[sa], sn[sb]
    ok = (d != 0.0) & (ia != NO_NODE) & (ib != NO_NODE) & (sa != sb)
    ia, ib, d = ia[ok], ib[ok], d[ok]
    if dims == 0:
        ld = np.abs(coords[ia] - coords[ib])
    else:
        c = coords.reshape(g.n_nodes, 2, dims)[:, 0, :]                # '+' end, sgd.rs:1268-1270
        ld = np.sqrt(((c[ia] - c[ib]) ** 2).sum(axis=1))
    return ((ld - d) / d) ** 2, ok


def sampled_stress(g, coords, dims=0, samples=10000, seed=12345):
    """sqrt(mean(((d_layout - d_path)/d_path)^2)) over uniform (step a, rank b) pairs — sgd.rs:1226-1282.
    dims = 0: coords is a 1D position vector x[dense index]; else Layout.coords order."""
    if g.n_steps < 2:
        return 0.0
    rng = np.random.default_rng(seed)
    pos, _ = g.step_positions()
    first = g.path_first_step.astype(np.int64)
    sa = rng.integers(0, g.n_steps, size=samples)
    p = np.searchsorted(first, sa, side="right") - 1
    cnt = first[p + 1] - first[p]
    sb = first[p] + (rng.random(samples) * cnt).astype(np.int64)
    keep = cnt >= 2
    e2, _ = _pair_errors(g, np.asarray(coords, dtype=np.float64), dims, sa[keep], sb[keep], pos)
    return float(np.sqrt(e2.mean())) if e2.size else 0.0


def step_distance_errors(g, coords, dims=0, z=1, pos=None):
    """Squared relative errors of ALL pairs of path steps (a, a + z) — no sampling.  The relative error at short path
    distances is heavy-tailed (a 1-bp node that sits 25 bp from its neighbour contributes 600 to a mean square of 0.04): a
    sample of 50 000 of the 1e7 adjacent pairs of a 525k-node graph swings by +-7 % with whether it hits one of the worst
    ten (profiles/r03/d1_outliers.log), the exhaustive figure is stable to 0.5 % between runs and seeds."""
    if pos is None:
        pos, _ = g.step_positions()
    first = g.path_first_step.astype(np.int64)
    S = g.n_steps
    path_of = np.repeat(np.arange(g.n_paths), np.diff(first))
    sa = np.arange(0, max(S - z, 0), dtype=np.int64)
    sa = sa[path_of[sa] == path_of[sa + z]]
    e2, _ = _pair_errors(g, np.asarray(coords, dtype=np.float64), dims, sa, sa + z, pos)
    return e2


def short_range_error(g, coords, dims=0, zs=(1,)):
    """dict(rms, trimmed_rms (without the worst 0.1 %), median, pairs) over all step pairs at the step distances `zs`."""
    pos, _ = g.step_positions()
    e2 = np.concatenate([step_distance_errors(g, coords, dims, z, pos) for z in zs])
    if e2.size == 0:
        return dict(rms=0.0, trimmed_rms=0.0, median=0.0, pairs=0)
    srt = np.sort(e2)
    return dict(rms=float(np.sqrt(srt.mean())), trimmed_rms=float(np.sqrt(srt[: max(1, int(srt.size * 0.999))].mean())),
                median=float(np.sqrt(srt[srt.size // 2])), pairs=int(srt.size))


def stress_by_scale(g, coords, dims=0, samples=400000, seed=777, max_octaves=32, exact_octaves=2):
    """Relative error by path distance: step a uniform, step distance 2^U with U uniform over [0, log2(path steps)).
    Returns (edges, rms_rel_err[octave], count[octave]): octave k holds step distances in [2^k, 2^(k+1)).
    The first `exact_octaves` octaves (step distances 1 and 2-3) are computed over ALL pairs instead of the sample
    (step_distance_errors says why)."""
    rng = np.random.default_rng(seed)
    pos, _ = g.step_positions()
    first = g.path_first_step.astype(np.int64)
    sa = rng.integers(0, g.n_steps, size=samples)
    p = np.searchsorted(first, sa, side="right") - 1
    cnt = first[p + 1] - first[p]
    keep = cnt >= 2
    sa, p, cnt = sa[keep], p[keep], cnt[keep]
    z = np.floor(np.exp2(rng.random(sa.shape[0]) * np.log2(cnt.astype(np.float64)))).astype(np.int64)
    z = np.maximum(z, 1)
    sgn = np.where(rng.random(sa.shape[0]) < 0.5, -1, 1)
    sb = sa + sgn * z
    bad = (sb < first[p]) | (sb >= first[p + 1])
    sb = np.where(bad, sa - sgn * z, sb)
    ok = (sb >= first[p]) & (sb < first[p + 1])
    sa, sb, z = sa[ok], sb[ok], z[ok]
    e2, used = _pair_errors(g, np.asarray(coords, dtype=np.float64), dims, sa, sb, pos)
    octv = np.floor(np.log2(z[used])).astype(np.int64)
    n_oct = int(min(max_octaves, octv.max() + 1)) if octv.size else 0
    rms = np.zeros(n_oct)
    num = np.zeros(n_oct, dtype=np.int64)
    for k in range(n_oct):
        if k < exact_octaves:
            ex = np.concatenate([step_distance_errors(g, coords, dims, zz, pos) for zz in range(1 << k, 2 << k)])
            num[k] = int(ex.size)
            rms[k] = float(np.sqrt(ex.mean())) if ex.size else 0.0
            continue
        m = octv == k
        num[k] = int(m.sum())
        rms[k] = float(np.sqrt(e2[m].mean())) if num[k] else 0.0
    return np.exp2(np.arange(n_oct + 1)), rms, num


# ---- rank agreement ------------------------------------------------------------------------------
def ranks_of(order):
    order = np.asarray(order, dtype=np.int64)
    r = np.empty(order.shape[0], dtype=np.int64)
    r[order] = np.arange(order.shape[0])
    return r


def _count_inversions(a):
    """Number of pairs i < j with a[i] > a[j]; a is a permutation of 0..n-1.  Bottom-up merge counting, O(n log n)
    numpy passes (a Fenwick tree in Python would take minutes at 2e6 nodes)."""
    a = np.asarray(a, dtype=np.int64).copy()
    n = a.shape[0]
    inv = 0
    width = 1
    idx = np.arange(n, dtype=np.int64)
    while width < n:
        # within every block of 2*width, count for each element of the right half how many left-half elements exceed it
        blk = idx // (2 * width)
        left = (idx % (2 * width)) < width
        # rank of each element inside its block after sorting = position by (blk, value)
        o = np.lexsort((a, blk))
        a_sorted_pos = np.empty(n, dtype=np.int64)
        a_sorted_pos[o] = idx
        # for right-half element e: (#elements of its block smaller than e) = rank in block; of those, the ones from
        # the right half are its rank within the (already sorted) right half
        blk_start = blk * (2 * width)
        rank_in_block = a_sorted_pos - blk_start
        right = ~left
        rank_in_right = (idx - blk_start - width)[right]           # halves are sorted: position = rank
        n_left = np.minimum(width, n - blk_start)[right]
        smaller_left = rank_in_block[right] - rank_in_right
        inv += int((n_left - smaller_left).sum())
        a = a[o]                                                    # blocks of 2*width are now sorted
        width *= 2
    return inv


def kendall_tau(rank_a, rank_b):
    """Kendall tau-a of two rankings without ties (permutations)."""
    rank_a = np.asarray(rank_a, dtype=np.int64)
    rank_b = np.asarray(rank_b, dtype=np.int64)
    n = rank_a.shape[0]
    if n < 2:
        return 1.0
    seq = rank_b[np.argsort(rank_a, kind="stable")]
    inv = _count_inversions(seq)
    tot = n * (n - 1) // 2
    return 1.0 - 2.0 * inv / tot


def spearman_rho(rank_a, rank_b):
    a = np.asarray(rank_a, dtype=np.float64)
    b = np.asarray(rank_b, dtype=np.float64)
    a -= a.mean()
    b -= b.mean()
    den = np.sqrt((a * a).sum() * (b * b).sum())
    return float((a * b).sum() / den) if den > 0 else 1.0


def oriented(rank_ref, rank):
    """A 1D layout is reflection-invariant: return `rank` or its mirror, whichever agrees with rank_ref."""
    rank = np.asarray(rank, dtype=np.int64)
    return rank if spearman_rho(rank_ref, rank) >= 0 else (rank.shape[0] - 1 - rank)


def inversions_vs_chain(node_ids_in_rank_order):
    """Adjacent inversions of a sort against a chain whose ids follow the chain (synthetic P1 graphs); mirror-invariant."""
    ids = np.asarray(node_ids_in_rank_order, dtype=np.int64)
    d = np.diff(ids)
    return int(min((d != 1).sum(), (d != -1).sum()))
