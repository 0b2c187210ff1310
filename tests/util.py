"""Shared helpers for the parity tests: oracle <-> product parameter plumbing."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402  (tests are allowed to use the oracle)
from gfasort_amd import graph as G  # noqa: E402
from gfasort_amd import params as P  # noqa: E402

DATA = os.path.join(ROOT, "tests", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")

_FIELDS = ["iter_max", "iter_with_max_learning_rate", "min_term_updates", "delta", "eps", "eta_max", "theta",
           "space", "space_max", "space_quantization_step", "cooling_start", "nthreads", "seed"]


def oracle_params(p, dimensions=2):
    kw = {k: getattr(p, k) for k in _FIELDS}
    kw["dimensions"] = getattr(p, "dimensions", dimensions)
    return O.params(**kw)


def oracle_graph(g):
    return O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)


def load(name):
    return G.load_gfa(os.path.join(DATA, name))


def gaussian_init(g, dims, seed):
    """Caller-side init of layout dims >= 1 (the reference uses rand_distr StandardNormal, which is
    not restated): Box-Muller on SplitMix64(seed), scaled by sqrt(2N) like sgd.rs:836."""
    n = g.n_nodes * 2 * dims
    r = G.splitmix64_array(seed, 2 * n)
    u1 = ((r[:n] >> np.uint64(11)).astype(np.float64) + 1.0) / 9007199254740993.0
    u2 = (r[n:] >> np.uint64(11)).astype(np.float64) / 9007199254740992.0
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    c = (z * np.sqrt(2.0 * g.n_nodes)).reshape(g.n_nodes, 2, dims)
    og = oracle_graph(g)
    c0 = O.init_layout_dim0(og, dims).reshape(g.n_nodes, 2, dims)
    c[:, :, 0] = c0[:, :, 0]
    return np.ascontiguousarray(c.reshape(-1))
