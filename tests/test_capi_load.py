"""The C-ABI library on a machine WITHOUT a GPU: it loads, exports every symbol the header
declares, its host tables equal the oracle bit for bit, and every compute entry point fails
loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from util import O, G, P, ROOT, load, oracle_graph, oracle_params
from gfasort_amd import hip


def _declared_functions():
    with open(os.path.join(ROOT, "include", "gfasort_hip.h")) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gfs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = hip.lib()
    declared = _declared_functions()
    assert len(declared) >= 25
    missing = [name for name in declared if not hasattr(L, name)]
    assert not missing, missing
    assert sorted(hip.EXPORTS) == declared
    assert b"gfx950" in L.gfs_version()


def test_struct_layouts_match_header():
    # sizes the header implies (natural alignment): the ctypes mirrors must agree
    assert C.sizeof(hip.SgdParams) == 14 * 8
    assert C.sizeof(hip.LayoutParams) == 15 * 8
    assert C.sizeof(hip.GraphView) == 7 * 8
    assert C.sizeof(hip.LaunchConfig) == 4 * 8 + 8 + 8
    assert C.sizeof(hip.Stats) == 9 * 8
    assert hip.TERM_DTYPE.itemsize == 16


@pytest.mark.parametrize("name", ["simple.gfa", "lil.gfa", "DRB1-3123.gfa"])
def test_host_tables_equal_oracle(name):
    g = load(name)
    for p in (P.YgsParams.from_graph(g, 0, 1).path_sgd, P.LayoutSGDParams.from_graph(g, 2, 1)):
        op = oracle_params(p)
        assert np.array_equal(hip.sgd_schedule(p).view(np.uint64), O.schedule(op).view(np.uint64))
        assert np.array_equal(hip.zeta_table(p).view(np.uint64), O.zetas(op).view(np.uint64))
    assert np.array_equal(hip.init_positions(g), O.init_positions(oracle_graph(g)))
    assert np.array_equal(hip.init_layout_dim0(g, 3), O.init_layout_dim0(oracle_graph(g), 3))


def test_fast_precise_pow_equals_oracle():
    rng = np.random.default_rng(3)
    for a in list(rng.uniform(1e-9, 3.0, 300)) + [0.5, 1.0, 2.0 / 156250]:
        for b in (0.99, 0.01, 0.001, 0.999, 100.00000000000009, 1.001001001001001):
            assert hip.fast_precise_pow(float(a), b) == O.fast_precise_pow(float(a), b)


def test_sort_order_ties_and_nan():
    x = np.array([3.0, 1.0, 2.0, 1.0, 0.5])
    assert hip.sort_order(x).tolist() == [4, 1, 3, 2, 0]              # stable: ties keep node_order
    assert hip.sort_order(np.zeros(0)).tolist() == []
    big = np.random.default_rng(0).normal(size=10000)
    assert np.array_equal(hip.sort_order(big), np.argsort(big, kind="stable").astype(np.uint64))


@pytest.mark.skipif(hip.lib().gfs_device_count() > 0, reason="a GPU is present")
def test_compute_entry_points_fail_loudly_without_gpu():
    g = load("simple.gfa")
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    with pytest.raises(hip.GfsError) as ei:
        hip.path_linear_sgd_raw(g, p)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)
    with pytest.raises(hip.GfsError):
        hip.Context(g)
    with pytest.raises(hip.GfsError):
        hip.path_linear_sgd_layout_raw(g, P.LayoutSGDParams.from_graph(g, 2, 1), np.zeros(g.n_nodes * 4))


def test_empty_graph_is_nothing_to_do_before_any_device_call():
    g = G.parse_gfa("H\tVN:Z:1.0\n")
    rc, x, st = hip.path_linear_sgd_raw(g, P.PathSGDParams())
    assert rc == hip.NOTHING_TO_DO and x.shape[0] == 0                # sgd.rs:242-244


def test_argument_validation():
    L = hip.lib()
    assert L.gfs_sgd_schedule(None, None) == -1 and b"null" in L.gfs_last_error()
    bad = hip.GraphView(1, 2, 1, None, None, None, None)
    h = C.c_void_p()
    assert L.gfs_ctx_create(C.byref(bad), 0, C.byref(h)) == -1


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gfasort_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                with open(os.path.join(dirpath, f), errors="ignore") as fh:
                    src = fh.read()
                assert "gfs_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f
