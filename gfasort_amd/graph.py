"""Host-side graph model for the SGD hot path: the flattened SoA mirror of the reference's
`BidirectedGraph` that the C-ABI consumes, a CLI-compatible GFA reader/writer, and the
seeded synthetic generators of SURVEY.md §8(d).

Reference: `src/graph.rs:9-64` (Handle), `src/graph_ops.rs:10-16,613-623` (BidirectedGraph,
add_node/node_order), `src/bin/gfasort.rs:88-167` (parse_gfa), `src/graph_ops.rs:693-738`
(write_gfa), `src/graph_ops.rs:1939-2025` (apply_ordering).
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

NO_NODE = 0xFFFFFFFF


@dataclass
class FlatGraph:
    """What `src/sgd.rs` reads from a BidirectedGraph, flattened.

    Dense node index k = position in `node_order` (GFA S-line order, graph_ops.rs:613-623);
    `handle_to_idx[Handle::forward(node_ids[k])] = k` at sgd.rs:286-294.
    """
    node_len: np.ndarray            # uint32[N]  sequence length by dense idx
    step_node: np.ndarray           # uint32[S]  dense idx of each step's node (NO_NODE if absent)
    step_is_rev: np.ndarray         # uint8[S]   Handle::is_reverse
    path_first_step: np.ndarray     # uint64[P+1]
    node_ids: np.ndarray            # uint64[N]  node_order (original ids)
    path_names: List[str] = field(default_factory=list)
    # only kept for small graphs read from GFA text (needed to write a GFA back):
    sequences: Optional[List[bytes]] = None          # by dense idx
    edges: Optional[List[Tuple[int, bool, int, bool]]] = None   # (from_id, from_rev, to_id, to_rev)
    step_node_id: Optional[np.ndarray] = None        # uint64[S] original ids of steps (for absent nodes)
    shared_node_layout: Optional[np.ndarray] = None  # multi-GPU: internal node layout common to all ranks

    @property
    def n_nodes(self):
        return int(self.node_len.shape[0])

    @property
    def n_steps(self):
        return int(self.step_node.shape[0])

    @property
    def n_paths(self):
        return int(self.path_first_step.shape[0]) - 1

    def path_step_counts(self):
        return np.diff(self.path_first_step.astype(np.int64))

    def step_positions(self):
        """PathIndex step_to_position and per-path bp length (sgd.rs:41-62)."""
        lens = np.where(self.step_node == NO_NODE, 0,
                        self.node_len[np.minimum(self.step_node, max(self.n_nodes - 1, 0))]
                        if self.n_nodes else 0).astype(np.uint64)
        csum = np.concatenate([[0], np.cumsum(lens, dtype=np.uint64)])
        first = self.path_first_step.astype(np.int64)
        path_of_step = np.repeat(np.arange(self.n_paths), np.diff(first))
        pos = csum[:-1] - csum[first[:-1]][path_of_step] if self.n_steps else csum[:-1]
        plen = csum[first[1:]] - csum[first[:-1]]
        return pos.astype(np.uint64), plen.astype(np.uint64)


# ----------------------------------------------------------------------------------------------
# GFA text <-> FlatGraph, CLI-compatible (numeric ids kept) — src/bin/gfasort.rs:88-167
# ----------------------------------------------------------------------------------------------
def parse_gfa(text: str) -> FlatGraph:
    node_seq = {}
    node_order: List[int] = []
    lines = text.split("\n")
    lines = [ln[:-1] if ln.endswith("\r") else ln for ln in lines]     # str::lines strips \r\n
    for ln in lines:
        if ln.startswith("S"):
            parts = ln.split("\t")
            if len(parts) >= 3:
                nid = _parse_usize(parts[1], "node ID")
                if nid not in node_seq:
                    node_order.append(nid)                              # graph_ops.rs:619-621
                node_seq[nid] = parts[2].encode()
    edges = []
    seen = set()
    for ln in lines:
        if ln.startswith("L"):
            parts = ln.split("\t")
            if len(parts) >= 5:
                f = _parse_usize(parts[1], "from ID")
                t = _parse_usize(parts[3], "to ID")
                e = (f, parts[2] != "+", t, parts[4] != "+")
                comp = (t, not e[3], f, not e[1])                       # graph_ops.rs:626-637
                if e not in seen and comp not in seen:
                    seen.add(e)
                    edges.append(e)
    idx_of = {nid: k for k, nid in enumerate(node_order)}
    step_node, step_rev, step_id, first, names = [], [], [], [0], []
    for ln in lines:
        if ln.startswith("P"):
            parts = ln.split("\t")
            if len(parts) >= 3:
                names.append(parts[1])
                for s in parts[2].split(","):
                    s = s.strip()
                    if not s:
                        continue
                    nid = _parse_usize(s[:-1], "path node ID")
                    step_id.append(nid)
                    step_node.append(idx_of.get(nid, NO_NODE))
                    step_rev.append(0 if s[-1] == "+" else 1)
                first.append(len(step_node))
    return FlatGraph(
        node_len=np.array([len(node_seq[n]) for n in node_order], dtype=np.uint32),
        step_node=np.array(step_node, dtype=np.uint32),
        step_is_rev=np.array(step_rev, dtype=np.uint8),
        path_first_step=np.array(first, dtype=np.uint64),
        node_ids=np.array(node_order, dtype=np.uint64),
        path_names=names,
        sequences=[node_seq[n] for n in node_order],
        edges=edges,
        step_node_id=np.array(step_id, dtype=np.uint64),
    )


def _parse_usize(s: str, what: str) -> int:
    if not s or not s.isdigit() and not (s[0] == "+" and s[1:].isdigit()):
        raise ValueError(f"Failed to parse {what}: invalid digit found in string")
    return int(s)


def load_gfa(path: str) -> FlatGraph:
    with open(path, "r") as fh:
        return parse_gfa(fh.read())


def apply_ordering_ids(g: FlatGraph, order_idx: np.ndarray) -> dict:
    """old node id -> new 1-based id, for an ordering given as dense indices
    (graph_ops.rs:1953-1957: `old_to_new[handle.node_id()] = new_idx + 1`)."""
    return {int(g.node_ids[k]): r + 1 for r, k in enumerate(order_idx.tolist())}


def write_gfa_sorted(g: FlatGraph, order_idx: np.ndarray) -> str:
    """GFA text of the graph after `apply_ordering(order)` (graph_ops.rs:1939-2025) as
    `write_gfa` prints it (graph_ops.rs:693-738).  The reference iterates a HashSet for the
    L lines (random order); here they are emitted sorted — compare L lines as a set."""
    assert g.sequences is not None and g.edges is not None
    o2n = apply_ordering_ids(g, order_idx)
    out = ["H\tVN:Z:1.0"]
    for r, k in enumerate(order_idx.tolist()):
        out.append(f"S\t{r + 1}\t{g.sequences[k].decode(errors='replace')}")
    new_edges = set()
    for (f, fr, t, tr) in g.edges:
        if f in o2n and t in o2n:                                       # graph_ops.rs:1983-1994
            new_edges.add((o2n[f], fr, o2n[t], tr))
    for (f, fr, t, tr) in sorted(new_edges):
        out.append(f"L\t{f}\t{'-' if fr else '+'}\t{t}\t{'-' if tr else '+'}\t0M")
    first = g.path_first_step.astype(np.int64)
    for p, name in enumerate(g.path_names):
        steps = []
        for s in range(first[p], first[p + 1]):
            nid = int(g.step_node_id[s])
            nid = o2n.get(nid, nid)                                     # graph_ops.rs:2007-2011
            steps.append(f"{nid}{'-' if g.step_is_rev[s] else '+'}")
        out.append(f"P\t{name}\t{','.join(steps)}\t*")
    return "\n".join(out) + "\n"


# ----------------------------------------------------------------------------------------------
# Seeded synthetic graphs — SURVEY.md §8(d).  All randomness is SplitMix64 so that the C++
# driver and this module generate identical graphs.
# ----------------------------------------------------------------------------------------------
_G = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_array(seed: int, n: int) -> np.ndarray:
    """First n outputs of SplitMix64(seed), vectorised."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _G * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _node_lengths(n: int, seed: int) -> np.ndarray:
    """len_i ~ U{1..16}: 1 + top 4 bits of SplitMix64(seed) output i (node id i+1)."""
    return (1 + (splitmix64_array(seed, n) >> np.uint64(60))).astype(np.uint32)


def _block_shuffled_order(n: int, seed: int, block: int = 64) -> np.ndarray:
    """S-line emission order: Fisher-Yates inside consecutive blocks of `block` node ids.
    Draw for (block b, position i) is SplitMix64(seed) output number b*block + i; j = draw % (i+1);
    positions i = block-1 .. 1 are swapped with j in that order."""
    order = np.arange(n, dtype=np.int64)
    r = splitmix64_array(seed, n)
    nb = n // block
    if nb:
        perm = order[: nb * block].reshape(nb, block).copy()
        rr = r[: nb * block].reshape(nb, block)
        rows = np.arange(nb)
        for i in range(block - 1, 0, -1):
            j = (rr[:, i] % np.uint64(i + 1)).astype(np.int64)
            a = perm[rows, i].copy()
            perm[rows, i] = perm[rows, j]
            perm[rows, j] = a
        order[: nb * block] = perm.reshape(-1)
    tail = n - nb * block
    if tail > 1:
        perm = order[nb * block:].copy()
        rr = r[nb * block:]
        for i in range(tail - 1, 0, -1):
            j = int(rr[i] % np.uint64(i + 1))
            perm[i], perm[j] = perm[j], perm[i]
        order[nb * block:] = perm
    return order          # order[k] = 0-based node number emitted k-th  (node id = order[k]+1)


def synth_windows(n_nodes: int, n_paths: int, window: int, seed: int, shuffle: bool = True) -> FlatGraph:
    """`windows(N,P,W,seed)`: nodes 1..N with len ~ U{1..16}; path p covers nodes
    o_p+1 .. o_p+W all forward, o_p = floor(p*(N-W)/(P-1)) (0 if P==1); S lines block-shuffled."""
    lens_by_id = _node_lengths(n_nodes, seed)                 # index = node id - 1
    order = _block_shuffled_order(n_nodes, seed + 1) if shuffle else np.arange(n_nodes, dtype=np.int64)
    inv = np.empty(n_nodes, dtype=np.int64)
    inv[order] = np.arange(n_nodes)
    offs = np.array([(p * (n_nodes - window)) // (n_paths - 1) if n_paths > 1 else 0
                     for p in range(n_paths)], dtype=np.int64)
    steps0 = (offs[:, None] + np.arange(window, dtype=np.int64)[None, :]).reshape(-1)   # 0-based node
    return FlatGraph(
        node_len=lens_by_id[order].astype(np.uint32),
        step_node=inv[steps0].astype(np.uint32),
        step_is_rev=np.zeros(steps0.shape[0], dtype=np.uint8),
        path_first_step=(np.arange(n_paths + 1, dtype=np.uint64) * np.uint64(window)),
        node_ids=(order + 1).astype(np.uint64),
        path_names=[f"p{p}" for p in range(n_paths)],
        step_node_id=(steps0 + 1).astype(np.uint64),
    )


def synth_bubbles(n_sites: int, n_haplotypes: int, seed: int, shuffle: bool = True) -> FlatGraph:
    """Pangenome-like graph: a backbone of n_sites "sites"; every 4th site is a biallelic bubble
    (two alternative nodes), every 16th site carries an optional insertion node, the others are a
    single shared node.  Each of n_haplotypes paths walks all sites choosing alleles at random
    (SplitMix64(seed+2)).  Node ids follow site order (allele 0, allele 1, insertion); S lines
    are block-shuffled like the other generators.  All steps forward."""
    site = np.arange(n_sites)
    is_bub = (site % 4 == 1)
    has_ins = (site % 16 == 7)
    n_per_site = 1 + is_bub.astype(np.int64) + has_ins.astype(np.int64)
    first_id0 = np.concatenate([[0], np.cumsum(n_per_site)])[:-1]            # 0-based id of the site's allele 0
    n_nodes = int(n_per_site.sum())
    lens_by_id = _node_lengths(n_nodes, seed)
    order = _block_shuffled_order(n_nodes, seed + 1) if shuffle else np.arange(n_nodes, dtype=np.int64)
    inv = np.empty(n_nodes, dtype=np.int64)
    inv[order] = np.arange(n_nodes)
    r = splitmix64_array(seed + 2, n_haplotypes * n_sites).reshape(n_haplotypes, n_sites)
    allele = ((r >> np.uint64(13)) & np.uint64(1)).astype(np.int64) * is_bub[None, :]
    take_ins = (((r >> np.uint64(29)) & np.uint64(3)) == 0) & has_ins[None, :]           # 25 % carry the insertion
    steps, firsts = [], [0]
    for h in range(n_haplotypes):
        main = first_id0 + allele[h]
        ins = first_id0 + 1 + is_bub.astype(np.int64)
        seq = np.stack([main, np.where(take_ins[h], ins, -1)], axis=1).reshape(-1)
        seq = seq[seq >= 0]
        steps.append(seq)
        firsts.append(firsts[-1] + seq.shape[0])
    steps0 = np.concatenate(steps)
    return FlatGraph(
        node_len=lens_by_id[order].astype(np.uint32),
        step_node=inv[steps0].astype(np.uint32),
        step_is_rev=np.zeros(steps0.shape[0], dtype=np.uint8),
        path_first_step=np.array(firsts, dtype=np.uint64),
        node_ids=(order + 1).astype(np.uint64),
        path_names=[f"h{h}" for h in range(n_haplotypes)],
        step_node_id=(steps0 + 1).astype(np.uint64),
    )


def tile_series(g: FlatGraph, copies: int, shuffle_seed: Optional[int] = None) -> FlatGraph:
    """A graph `copies` times in series: copy c holds the nodes of g with dense indices (and ids) offset by c * N, and every
    path of g becomes ONE path that walks copy 0, then copy 1, ... (orientations kept) — a path that traverses g mostly on
    the reverse strand walks the copies in DESCENDING order instead, as a reverse-strand haplotype of the tiled graph would
    (DRB1-3123's path 6 is reverse from end to end: walking the copies upwards would put an artificial inversion at every
    seam, and those 119 seams then carry 60 % of the squared error at path distance 1, profiles/r03/tiled_tail_probe_naive_seams.log).
    With a real fixture — the
    reference's tests/data/DRB1-3123.gfa: nested bubbles, 3 096 reverse steps — this gives a graph of real pangenome
    structure at the sizes where the library's default kernels differ from the reference's sampler.
    shuffle_seed: emit the S lines block-shuffled (Fisher-Yates inside blocks of 64 nodes) as the synthetic generators do,
    so that node_order is not the input's order."""
    N, S, P = g.n_nodes, g.n_steps, g.n_paths
    first = g.path_first_step.astype(np.int64)
    counts = np.diff(first)
    off = (np.arange(copies, dtype=np.int64) * N)
    steps, revs = [], []
    for p in range(P):
        seg = g.step_node[first[p]:first[p + 1]].astype(np.int64)
        absent = seg == NO_NODE
        rev_path = seg.size > 0 and 2 * int(g.step_is_rev[first[p]:first[p + 1]].sum()) > seg.size
        tiled = seg[None, :] + (off[::-1] if rev_path else off)[:, None]
        tiled[:, absent] = NO_NODE
        steps.append(tiled.reshape(-1))
        revs.append(np.tile(g.step_is_rev[first[p]:first[p + 1]], copies))
    steps0 = np.concatenate(steps) if steps else np.zeros(0, dtype=np.int64)
    n_total = N * copies
    order = _block_shuffled_order(n_total, shuffle_seed) if shuffle_seed is not None else np.arange(n_total, dtype=np.int64)
    inv = np.empty(n_total, dtype=np.int64)
    inv[order] = np.arange(n_total)
    node_len = np.tile(g.node_len, copies)
    max_id = int(g.node_ids.max()) if N else 0
    ids = (g.node_ids.astype(np.int64)[None, :] + (np.arange(copies, dtype=np.int64) * max_id)[:, None]).reshape(-1)
    sn = np.where(steps0 == NO_NODE, NO_NODE, inv[np.minimum(steps0, max(n_total - 1, 0))])
    return FlatGraph(
        node_len=node_len[order].astype(np.uint32),
        step_node=sn.astype(np.uint32),
        step_is_rev=(np.concatenate(revs) if revs else np.zeros(0)).astype(np.uint8),
        path_first_step=np.concatenate([[0], np.cumsum(counts * copies)]).astype(np.uint64),
        node_ids=ids[order].astype(np.uint64),
        path_names=list(g.path_names),
    )


def synth_chain(n_nodes: int, seed: int, shuffle: bool = True) -> FlatGraph:
    """`chain(N,seed)`: one path 1+,..,N+ over a linear chain; S lines block-shuffled."""
    return synth_windows(n_nodes, 1, n_nodes, seed, shuffle)


def synth_to_gfa_text(g: FlatGraph) -> str:
    """GFA text of a synthetic graph (sequence = 'A'*len; edges i+ -> (i+1)+)."""
    out = ["H\tVN:Z:1.0"]
    for k in range(g.n_nodes):
        out.append(f"S\t{int(g.node_ids[k])}\t{'A' * int(g.node_len[k])}")
    for i in range(1, g.n_nodes):
        out.append(f"L\t{i}\t+\t{i + 1}\t+\t0M")
    first = g.path_first_step.astype(np.int64)
    for p, name in enumerate(g.path_names):
        ids = g.step_node_id[first[p]:first[p + 1]]
        out.append(f"P\t{name}\t{','.join(f'{int(i)}+' for i in ids)}\t*")
    return "\n".join(out) + "\n"


def synth_repeats(n: int, n_paths: int, period: int, max_copies: int, every: int, seed: int) -> FlatGraph:
    """Chain of n nodes in which every `every`-th position starts a block of `period` nodes that each path
    traverses 1..max_copies times in a row (copy-number variation / tandem repeats; period 1 = self-loops).
    Consecutive path steps then revisit the same nodes, which a linear chain never does."""
    rng = np.random.default_rng(seed)
    node_len = rng.integers(1, 17, n).astype(np.uint32)
    steps, first = [], [0]
    for _ in range(n_paths):
        k = 0
        while k < n:
            if k % every == 0 and k + period <= n:
                c = int(rng.integers(1, max_copies + 1))
                for _ in range(c):
                    steps.extend(range(k, k + period))
                k += period
            else:
                steps.append(k)
                k += 1
        first.append(len(steps))
    return FlatGraph(node_len=node_len, step_node=np.array(steps, dtype=np.uint32),
                     step_is_rev=np.zeros(len(steps), dtype=np.uint8), path_first_step=np.array(first, dtype=np.uint64),
                     node_ids=np.arange(1, n + 1, dtype=np.uint64), path_names=[f"p{i}" for i in range(n_paths)])
