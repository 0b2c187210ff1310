"""Host-side mirror of the reference's SGD entry points, running on the HIP engine.

Reference call chain (src/ygs.rs:195-206 -> src/sgd.rs:641-672 -> src/sgd.rs:237-614):
    sgd_sort_only -> path_sgd_sort -> path_linear_sgd
and the layout arm (src/bin/gfasort.rs:265-274 -> src/sgd.rs:773-1188).
Same names, argument meaning and empty-result behaviour; positions are indexed by the dense
node index (position in node_order) exactly like the reference's HashMap<usize,f64> keys.
"""
from typing import Optional, Tuple

import numpy as np

from . import hip
from .graph import FlatGraph
from .layout import Layout
from .params import LayoutSGDParams, PathSGDParams


def path_linear_sgd(graph: FlatGraph, params: PathSGDParams, cfg=None,
                    return_stats: bool = False):
    """sgd.rs:237.  Returns float64[n_nodes] positions, or an EMPTY array when the reference
    returns an empty map (no nodes / no path with more than one step, sgd.rs:242-244,258-261)."""
    if graph.n_nodes == 0:
        return (np.zeros(0), None) if return_stats else np.zeros(0)
    rc, x, st = hip.path_linear_sgd_raw(graph, params, cfg=cfg)
    if rc == hip.NOTHING_TO_DO:
        x = np.zeros(0)
    return (x, st) if return_stats else x


def path_sgd_sort(graph: FlatGraph, params: PathSGDParams, cfg=None) -> np.ndarray:
    """sgd.rs:641-672: dense node indices in ascending position order.  (The reference breaks
    ties by HashMap iteration order, i.e. randomly; here ties keep node_order.)"""
    if graph.n_nodes == 0:
        return np.zeros(0, dtype=np.uint64)
    rc, x, order, st = hip.path_sgd_sort_raw(graph, params, cfg=cfg)       # SGD + device radix sort
    if rc == hip.NOTHING_TO_DO:
        return np.zeros(0, dtype=np.uint64)
    return order


def sgd_sort_only(graph: FlatGraph, params: PathSGDParams, verbose: int = 0, cfg=None) -> np.ndarray:
    """ygs.rs:195-206.  Returns the ordering that `apply_ordering` would consume (empty = the
    reference's no-op, graph_ops.rs:1940)."""
    return path_sgd_sort(graph, params, cfg)


def default_layout_init(graph: FlatGraph, dims: int, seed: int) -> np.ndarray:
    """Initial coordinates in Layout order, as the reference draws them (sgd.rs:829-853): dimension 0 = bp prefix
    (+ end) / prefix + length (- end); dimensions >= 1 = StandardNormal * sqrt(2N) from ONE Xoshiro256+ seeded `seed`,
    node by node.  gfs_init_layout restates rand_distr's ziggurat from its published algorithm: parity unpinned
    (DESIGN.md §5)."""
    return hip.init_layout(graph, dims, seed)


def path_linear_sgd_layout(graph: FlatGraph, params: LayoutSGDParams, init: Optional[np.ndarray] = None,
                           cfg=None, return_stats: bool = False):
    """sgd.rs:773.  Returns a Layout; all-zero when the reference returns `Layout::new`
    (sgd.rs:780-782,795-798)."""
    D = params.dimensions
    if graph.n_nodes == 0:
        lay = Layout(D, 0)
        return (lay, None) if return_stats else lay
    if init is None:
        init = default_layout_init(graph, D, params.seed)
    rc, coords, st = hip.path_linear_sgd_layout_raw(graph, params, init, cfg=cfg)
    lay = Layout(D, graph.n_nodes, coords if rc == hip.OK else None)
    return (lay, st) if return_stats else lay
