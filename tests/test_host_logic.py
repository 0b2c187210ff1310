"""Host-side logic (CPU): GFA reader/writer, flattening, Layout, synthetic generators, sharding.
Mirrors the reference's own unit tests where they exist (layout.rs:262-340, ygs.rs:247-303,
graph_ops.rs:2051-2132, integration_tests.rs)."""
import io

import numpy as np
import pytest

from util import O, G, P, load, oracle_graph
from gfasort_amd.layout import Layout, _fmt_f64, dim_name
from gfasort_amd import distributed as D


# ---- GFA reader (src/bin/gfasort.rs:88-167) --------------------------------------------------------
def test_parse_gfa_simple():
    g = load("simple.gfa")
    assert g.node_ids.tolist() == list(range(1, 16))                  # S-line order = node_order
    assert g.node_len[:3].tolist() == [8, 1, 1] and g.node_len[8] == 19
    assert g.path_names == ["x"] and g.step_node.tolist() == [0, 2, 4, 5, 7, 8, 10, 11, 13, 14]
    assert len(g.edges) == 20 and not g.step_is_rev.any()


def test_parse_gfa_node_order_duplicates_and_absent_nodes():
    txt = "S\t7\tACG\nS\t3\tA\nS\t7\tTTTT\nP\tp\t3+,7-,9+\t*\nL\t7\t+\t3\t-\t0M\nL\t3\t+\t7\t-\t0M\n"
    g = G.parse_gfa(txt)
    assert g.node_ids.tolist() == [7, 3]                              # 7 pushed once (graph_ops.rs:619-621)
    assert g.node_len.tolist() == [4, 1]                              # sequence overwritten by the 2nd S line
    assert g.step_node.tolist() == [1, 0, G.NO_NODE]                  # 9 is not a node
    assert g.step_is_rev.tolist() == [0, 1, 0]
    assert len(g.edges) == 1                                          # 3+->7- is the complement of 7+->3-
    pos, plen = g.step_positions()
    assert pos.tolist() == [0, 1, 5] and plen.tolist() == [5]         # absent node adds 0 bp (sgd.rs:52-54)


def test_parse_gfa_rejects_bad_ids():
    with pytest.raises(ValueError):
        G.parse_gfa("S\tabc\tA\n")


def test_write_gfa_sorted_roundtrip_counts():
    # integration_tests.rs:23-51,175-206: sorting keeps node / edge / path counts
    g = load("DRB1-3123.gfa")
    order = np.random.default_rng(0).permutation(g.n_nodes)
    txt = G.write_gfa_sorted(g, order)
    g2 = G.parse_gfa(txt)
    assert (g2.n_nodes, len(g2.edges), g2.n_paths, g2.n_steps) == (g.n_nodes, len(g.edges), g.n_paths, g.n_steps)
    assert g2.node_ids.tolist() == list(range(1, g.n_nodes + 1))      # new id = rank+1 (graph_ops.rs:1956)
    assert txt.startswith("H\tVN:Z:1.0\n")
    # the node at rank r carries the sequence of the node that was ordered r-th
    assert g2.sequences[5] == g.sequences[int(order[5])]
    # path step sequences are preserved
    seq = lambda gg, s: gg.sequences[int(gg.step_node[s])]
    assert all(seq(g, s) == seq(g2, s) for s in range(0, g.n_steps, 97))
    assert np.array_equal(g.step_is_rev, g2.step_is_rev)


def test_apply_ordering_ids():
    g = load("simple.gfa")
    o2n = G.apply_ordering_ids(g, np.array([2, 0, 1]))
    assert o2n == {3: 1, 1: 2, 2: 3}


# ---- Layout (src/layout.rs:258-341) ------------------------------------------------------------------
def test_layout_new_get_set():
    lay = Layout(2, 10)
    assert (lay.dimensions, lay.num_nodes, len(lay.coords)) == (2, 10, 40)
    lay = Layout(2, 5)
    lay.set(2, 0, 0, 100.0); lay.set(2, 0, 1, 200.0); lay.set(2, 1, 0, 150.0); lay.set(2, 1, 1, 250.0)
    assert (lay.x_plus(2), lay.y_plus(2), lay.x_minus(2), lay.y_minus(2)) == (100.0, 200.0, 150.0, 250.0)
    assert lay.index(2, 1, 1) == 2 * 2 * 2 + 1 * 2 + 1


def test_layout_distance_345():
    lay = Layout(2, 2)
    lay.set(1, 0, 0, 3.0); lay.set(1, 0, 1, 4.0)
    assert abs(lay.distance(0, 0, 1, 0) - 5.0) < 1e-10


def test_layout_from_vectors():
    lay = Layout.from_vectors([np.array([1.0, 2.0, 3.0, 4.0]), np.array([10.0, 20.0, 30.0, 40.0])])
    assert (lay.num_nodes, lay.dimensions) == (2, 2)
    assert (lay.x_plus(0), lay.y_plus(0), lay.x_minus(0), lay.y_minus(0), lay.x_plus(1), lay.y_plus(1)) == \
        (1.0, 10.0, 2.0, 20.0, 3.0, 30.0)


def test_layout_tsv_roundtrip():
    lay = Layout(2, 3)
    vals = [1.5, 2.5, 3.5, 4.5, 10.0, 20.0, 30.0, 40.0, 100.0, 200.0, 300.0, 400.0]
    lay.coords[:] = vals
    txt = lay.to_tsv()
    assert txt.split("\n")[0] == "idx\tx+\ty+\tx-\ty-"
    assert txt.split("\n")[2] == "1\t10\t20\t30\t40"                   # Rust `{}` prints 10.0 as "10"
    back = Layout.read_tsv(io.StringIO(txt))
    assert (back.dimensions, back.num_nodes) == (2, 3) and np.allclose(back.coords, lay.coords, atol=1e-10)


def test_rust_float_display():
    assert _fmt_f64(1.0) == "1" and _fmt_f64(-0.5) == "-0.5" and _fmt_f64(1e21) == "1000000000000000000000"
    assert _fmt_f64(1e-7) == "0.0000001" and _fmt_f64(0.1 + 0.2) == "0.30000000000000004"
    assert [dim_name(d) for d in range(6)] == ["x", "y", "z", "w", "d", "d"]


# ---- synthetic generators --------------------------------------------------------------
def test_synth_chain_structure():
    g = G.synth_chain(1000, 1)
    assert (g.n_nodes, g.n_steps, g.n_paths) == (1000, 1000, 1)
    assert sorted(g.node_ids.tolist()) == list(range(1, 1001))
    assert g.node_ids.tolist() != list(range(1, 1001))                # S lines are block-shuffled
    assert g.step_node_id.tolist() == list(range(1, 1001))            # the path is the chain
    assert g.node_ids[g.step_node].tolist() == list(range(1, 1001))
    assert 1 <= g.node_len.min() and g.node_len.max() <= 16
    # shuffle stays inside blocks of 64 ids
    assert all(abs(int(nid) - 1 - k) < 64 and (int(nid) - 1) // 64 == k // 64 for k, nid in enumerate(g.node_ids))
    g2 = G.synth_chain(1000, 1)
    assert np.array_equal(g.node_ids, g2.node_ids) and np.array_equal(g.node_len, g2.node_len)


def test_synth_windows_c3_shape_small():
    g = G.synth_windows(10000, 8, 2000, 2)
    assert (g.n_nodes, g.n_paths, g.n_steps) == (10000, 8, 16000)
    first = g.path_first_step.astype(int)
    for p in range(8):
        ids = g.step_node_id[first[p]:first[p + 1]]
        o = (p * (10000 - 2000)) // 7
        assert ids[0] == o + 1 and ids[-1] == o + 2000 and np.all(np.diff(ids.astype(int)) == 1)
    y = P.YgsParams.from_graph(g, 0, 1).path_sgd
    assert y.min_term_updates == 16000 and y.eta_max == 2000.0 ** 2


def test_synth_gfa_text_roundtrip():
    g = G.synth_windows(300, 3, 120, 5)
    g2 = G.parse_gfa(G.synth_to_gfa_text(g))
    assert np.array_equal(g.node_ids, g2.node_ids) and np.array_equal(g.node_len, g2.node_len)
    assert np.array_equal(g.step_node, g2.step_node) and np.array_equal(g.path_first_step, g2.path_first_step)


def test_splitmix_array_matches_oracle():
    assert G.splitmix64_array(1234567, 5).tolist() == O.splitmix64_stream(1234567, 5)


# ---- multi-GPU sharding helpers ----------------------------------------------------------------------------
def _graph_with_counts(counts, n_nodes=64):
    """A FlatGraph whose paths have the given step counts (steps walk the nodes cyclically)."""
    steps = np.concatenate([np.arange(c) % n_nodes for c in counts]).astype(np.uint32) if len(counts) else np.zeros(0, np.uint32)
    return G.FlatGraph(node_len=np.ones(n_nodes, dtype=np.uint32), step_node=steps, step_is_rev=np.zeros(steps.shape[0], np.uint8),
                       path_first_step=np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64),
                       node_ids=np.arange(1, n_nodes + 1, dtype=np.uint64), path_names=[f"p{k}" for k in range(len(counts))])


def test_shard_paths_balanced_and_complete():
    """gfs_shard_paths (host-only part of the multi-device ABI): every path has one owner, loads are balanced,
    one-step paths weigh nothing, and blocks are kept when they are balanced."""
    from gfasort_amd import hip
    counts = [100, 90, 80, 10, 10, 10, 5, 5, 1, 1]
    g = _graph_with_counts(counts)
    plan = hip.ShardPlan(g, 1000, 3)
    assert sorted(p for r in range(3) for p in plan.paths_of(r)) == list(range(10))
    loads = [sum(counts[p] for p in plan.paths_of(r) if counts[p] > 1) for r in range(3)]
    assert loads == [int(v) for v in plan.rank_steps] and max(loads) - min(loads) <= 20      # unbalanced blocks -> bin packing
    assert int(plan.quotas.sum()) == 1000
    assert hip.ShardPlan(g, 1000, 1).paths_of(0) == list(range(10))
    even = _graph_with_counts([50] * 8)
    plan = hip.ShardPlan(even, 800, 4)
    assert [plan.paths_of(r) for r in range(4)] == [[0, 1], [2, 3], [4, 5], [6, 7]]            # consecutive blocks
    assert [int(q) for q in plan.quotas] == [200] * 4
    # more ranks than paths: the surplus ranks own nothing and get no updates
    plan = hip.ShardPlan(_graph_with_counts([10, 10, 10]), 30, 4)
    assert sorted(int(q) for q in plan.quotas) == [0, 10, 10, 10] and int(plan.quotas.sum()) == 30


def test_exchange_plan_spans_and_shared_slots():
    """gfs_exchange_plan: windows over a chain — each rank's paths touch one span of the shared layout; only the
    overlaps are exchanged; every slot has exactly one designated owner (the lowest covering rank; rank 0 for slots no
    span covers, so that gfs_rank_finish_* leaves untouched nodes where they started — sgd.rs:286-294)."""
    from gfasort_amd import hip
    g = G.synth_windows(10_000, 8, 2_000, 3, shuffle=False)            # path p covers nodes o_p .. o_p + 2000
    plan = hip.ShardPlan(g, 1000, 4)
    assert np.array_equal(plan.perm, np.arange(g.n_nodes))               # already in path order
    first = g.path_first_step.astype(int)
    for r in range(4):
        nodes = np.concatenate([g.step_node[first[p]:first[p + 1]] for p in plan.paths_of(r)])
        assert (int(plan.span_lo[r]), int(plan.span_hi[r])) == (int(nodes.min()), int(nodes.max()) + 1)
    cover = np.zeros(g.n_nodes, dtype=int)
    for r in range(4):
        cover[int(plan.span_lo[r]):int(plan.span_hi[r])] += 1
    shared = np.zeros(g.n_nodes, dtype=bool)
    for lo, hi in plan.shared:
        shared[lo:hi] = True
    assert np.array_equal(shared, cover >= 2) and 0 < shared.sum() < g.n_nodes
    owner = np.full(g.n_nodes, -1)
    for lo, hi, r in plan.owned:
        assert (owner[lo:hi] == -1).all()
        owner[lo:hi] = r
    assert (owner >= 0).all()                                            # the intervals tile [0, n_nodes)
    assert (owner[cover == 0] == 0).all()
    for r in range(4):
        mine = (owner == r) & (cover >= 1)
        assert (mine[:int(plan.span_lo[r])] == False).all() and (mine[int(plan.span_hi[r]):] == False).all()
    # gaps: paths over two far-apart stretches of a chain with unvisited nodes before, between and after them
    n = 1000
    steps = [np.arange(100, 300), np.arange(250, 400), np.arange(600, 800), np.arange(700, 900)]
    counts = [len(v) for v in steps]
    gg = G.FlatGraph(node_len=np.full(n, 2, dtype=np.uint32), step_node=np.concatenate(steps).astype(np.uint32),
                     step_is_rev=np.zeros(sum(counts), np.uint8),
                     path_first_step=np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64),
                     node_ids=np.arange(1, n + 1, dtype=np.uint64), path_names=[f"p{k}" for k in range(4)])
    pl = hip.ShardPlan(gg, 750, 4)
    owner = np.full(n, -1)
    for lo, hi, r in pl.owned:
        assert (owner[lo:hi] == -1).all()
        owner[lo:hi] = r
    assert (owner >= 0).all()
    slot_of = pl.perm.astype(int)                                        # visited nodes first, unvisited last
    unvisited = np.setdiff1d(np.arange(n), np.concatenate(steps))
    assert (owner[slot_of[unvisited]] == 0).all()
    assert len(pl.owned) <= 2 * 4 + 1
    whole = hip.ShardPlan(g, 1000, 4, whole_vector=True)
    assert whole.shared == [(0, g.n_nodes)]


def test_shard_quotas_sum_exactly():
    for total, steps in [(10_000_000, [1250000] * 8), (35059, [17000, 18059]), (7, [1, 1, 1]), (5, [0, 10])]:
        q = D.shard_quotas(total, steps)
        assert sum(q) == total and all(v >= 0 for v in q)
    assert D.shard_quotas(100, [0, 10]) == [0, 100]


def test_subgraph_keeps_nodes_and_selected_paths():
    g = load("DRB1-3123.gfa")
    sub = D.subgraph(g, [1, 4])
    counts = g.path_step_counts()
    assert sub.n_nodes == g.n_nodes and sub.n_paths == 2 and sub.n_steps == counts[1] + counts[4]
    f = g.path_first_step.astype(int)
    assert np.array_equal(sub.step_node[:counts[1]], g.step_node[f[1]:f[2]])


def _layout_reference(g):
    """The node layout rule, restated with plain loops (gfs_shared_node_layout / index_kernels.hip): first-visit path order,
    except that a run of first visits which does not start its path goes right after the root-run node it branches off."""
    N, S = g.n_nodes, g.n_steps
    node = g.step_node
    starts = set(int(v) for v in g.path_first_step[:-1])
    first = {}
    for s in range(S):
        n = int(node[s])
        if n != 0xFFFFFFFF and n not in first:
            first[n] = s
    root = list(range(N))
    prev_first, anchor = False, None
    for s in range(S):
        if s in starts:
            prev_first = False
        n = int(node[s])
        is_first = n != 0xFFFFFFFF and first[n] == s
        if is_first:
            if not prev_first:
                a = None if s in starts else int(node[s - 1])
                anchor = None if a in (None, 0xFFFFFFFF) else a
            if anchor is not None:
                root[n] = root[anchor]
        prev_first = is_first
    big = 1 << 62
    def key(k):
        if k not in first:
            return (big, big, k)
        return (2 * first[k] if root[k] == k else 2 * first[root[k]] + 1, first[k], k)
    order = sorted(range(N), key=key)
    perm = np.empty(N, dtype=np.uint32)
    perm[order] = np.arange(N, dtype=np.uint32)
    return perm


def test_shared_node_layout_places_branches_where_they_branch_off():
    from gfasort_amd.distributed import path_order_layout
    rng = np.random.default_rng(4)
    n, S = 400, 3000
    step_node = rng.integers(0, n // 2, S).astype(np.uint32) * 2
    step_node[rng.integers(0, S, 40)] = 0xFFFFFFFF
    messy = G.FlatGraph(node_len=np.ones(n, np.uint32), step_node=step_node, step_is_rev=np.zeros(S, np.uint8),
                        path_first_step=np.array([0, 0, S // 3, S // 3, S], dtype=np.uint64), node_ids=np.arange(1, n + 1, dtype=np.uint64),
                        path_names=["a", "b", "c", "d"])
    bub = G.synth_bubbles(3000, 6, 5)
    for g in (load("lil.gfa"), load("DRB1-3123.gfa"), messy, bub, G.synth_windows(2000, 4, 1000, 3)):
        perm = path_order_layout(g)
        assert sorted(perm.tolist()) == list(range(g.n_nodes))
        assert np.array_equal(perm, _layout_reference(g))
    # what it is for: consecutive path steps of a bubble graph sit next to each other in the layout — with plain first-visit
    # order a fifth of them were thousands of slots apart (the alternative alleles)
    slots = path_order_layout(bub)[bub.step_node].astype(np.int64)
    inside = np.ones(bub.n_steps - 1, dtype=bool)
    inside[(bub.path_first_step[1:-1] - 1).astype(np.int64)] = False
    far = np.abs(np.diff(slots))[inside] > 8
    assert far.mean() < 0.001, far.mean()
    # a graph without branches (every path a window of one chain) keeps plain first-visit order
    w = G.synth_windows(2000, 4, 1000, 3)
    seen, order = set(), []
    for v in w.step_node.tolist():
        if v not in seen:
            seen.add(v); order.append(v)
    assert np.array_equal(np.argsort(path_order_layout(w))[:len(order)], np.array(order))
