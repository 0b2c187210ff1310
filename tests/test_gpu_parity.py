"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

P0  replay: ONE stream on the GPU == oracle single stream, bit for bit (positions).
P0' sampler: the first k terms (i, j, d_ij) of EVERY stream at full width == oracle, bit for bit
    (sampling does not depend on positions, so this is exact at any width).
"""
import numpy as np
import pytest

from util import O, G, P, load, oracle_graph, oracle_params, gaussian_init
from gfasort_amd import hip

pytestmark = pytest.mark.gpu


def _ygs(g, iter_max):
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = iter_max
    return p


@pytest.mark.parametrize("name,iter_max", [("simple.gfa", 100), ("lil.gfa", 100), ("DRB1-3123.gfa", 10)])
def test_replay_1d_bitexact(name, iter_max):
    g = load(name)
    p = _ygs(g, iter_max)
    og, op = oracle_graph(g), oracle_params(p)
    x_ref = O.init_positions(og)
    rc, st, _ = O.sgd_1d(og, op, x_ref, n_streams=1)
    assert rc == 0
    cfg = hip.make_config(n_streams=1)
    rc, x, hst = hip.path_linear_sgd_raw(g, p, cfg=cfg)
    assert rc == 0
    assert hst.term_updates == st.term_updates == (iter_max + 1) * p.min_term_updates
    assert hst.attempts == st.attempts
    assert np.array_equal(x.view(np.uint64), x_ref.view(np.uint64))


@pytest.mark.parametrize("flags", [0, hip.F_PLAIN_LOADS, hip.F_NO_LDS_TABLES])
def test_replay_1d_variants(flags):
    g = load("DRB1-3123.gfa")
    p = _ygs(g, 4)
    og, op = oracle_graph(g), oracle_params(p)
    x_ref = O.init_positions(og)
    O.sgd_1d(og, op, x_ref, n_streams=1)
    rc, x, _ = hip.path_linear_sgd_raw(g, p, cfg=hip.make_config(n_streams=1, flags=flags))
    assert rc == 0
    assert np.array_equal(x.view(np.uint64), x_ref.view(np.uint64))


def test_sampler_trace_full_width_1d():
    g = load("DRB1-3123.gfa")
    p = _ygs(g, 6)     # crosses into the cooling half (k > 3)
    T, K = 4096, 40
    og, op = oracle_graph(g), oracle_params(p)
    x_ref = O.init_positions(og)
    rc, st, tr_ref = O.sgd_1d(og, op, x_ref, n_streams=T, trace_per_stream=K)
    ctx = hip.Context(g)
    cfg = hip.make_config(n_streams=T, trace_per_stream=K)
    assert ctx.setup_1d(p, cfg) == 0
    ctx.upload(hip.init_positions(g))
    ctx.run()
    tr, counts = ctx.trace()
    hst = ctx.stats()
    assert hst.term_updates == st.term_updates
    assert hst.attempts == st.attempts
    tr_ref = tr_ref.reshape(T, K)
    assert np.array_equal(tr["i"], tr_ref["i"])
    assert np.array_equal(tr["j"], tr_ref["j"])
    assert np.array_equal(tr["d_ij"].view(np.uint64), tr_ref["d_ij"].view(np.uint64))
    ctx.close()


@pytest.mark.parametrize("dims", [2, 3])
def test_replay_nd_bitexact(dims):
    g = load("DRB1-3123.gfa")
    p = P.LayoutSGDParams.from_graph(g, dims, 1)
    p.iter_max = 3
    p.min_term_updates = 20000
    og, op = oracle_graph(g), oracle_params(p)
    c0 = gaussian_init(g, dims, 7)
    c_ref = c0.copy()
    rc, st, _ = O.sgd_nd(og, op, c_ref, n_streams=1)
    assert rc == 0
    rc, c, hst = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(n_streams=1))
    assert rc == 0
    assert hst.term_updates == st.term_updates
    assert hst.attempts == st.attempts
    assert np.array_equal(c.view(np.uint64), c_ref.view(np.uint64))


def test_sampler_trace_full_width_nd():
    g = load("DRB1-3123.gfa")
    p = P.LayoutSGDParams.from_graph(g, 2, 1)
    p.iter_max = 4
    p.min_term_updates = 100000
    T, K = 2048, 64
    og, op = oracle_graph(g), oracle_params(p)
    c0 = gaussian_init(g, 2, 7)
    c_ref = c0.copy()
    rc, st, tr_ref = O.sgd_nd(og, op, c_ref, n_streams=T, trace_per_stream=K)
    ctx = hip.Context(g)
    assert ctx.setup_nd(p, hip.make_config(n_streams=T, trace_per_stream=K)) == 0
    ctx.upload(c0)
    ctx.run()
    tr, counts = ctx.trace()
    hst = ctx.stats()
    assert hst.term_updates == st.term_updates and hst.attempts == st.attempts
    tr_ref = tr_ref.reshape(T, K)
    assert np.array_equal(tr["i"], tr_ref["i"]) and np.array_equal(tr["j"], tr_ref["j"])
    assert np.array_equal(tr["d_ij"].view(np.uint64), tr_ref["d_ij"].view(np.uint64))
    ctx.close()
