"""Oracle known-answer tests (CPU).  Pins the oracle against (1) the published reference vectors
of xoshiro256+ / SplitMix64, (2) independent pure-Python restatements of the rand-0.9 samplers
and of fast_precise_pow, (3) values the reference's own files fix (fixture parameter table,
PathIndex of simple.gfa, schedule end points, defaults), (4) the committed golden file."""
import json
import math
import os
import struct

import numpy as np
import pytest

from util import O, G, P, GOLDEN, load, oracle_graph, oracle_params, gaussian_init

M64 = (1 << 64) - 1


# ---- (1) published generator vectors ------------------------------------------------------------
def test_xoshiro256plus_reference_vector():
    # xoshiro256+ from state {1,2,3,4}: outputs of Blackman & Vigna's reference C code
    # (the same vector rand_xoshiro's own unit test uses)
    r = O.Xoshiro(state=[1, 2, 3, 4])
    want = [5, 211106232532999, 211106635186183, 9223759065350669058, 9250833439874351877,
            13862484359527728515, 2346507365006083650, 1168864526675804870, 34095955243042024,
            3466914240207415127]
    assert [r.next_u64() for _ in range(10)] == want


def test_splitmix64_reference_vector():
    # splitmix64.c with x = 1234567
    assert O.splitmix64_stream(1234567, 5) == [6457827717110365317, 3203168211198807973,
                                              9817491932198370423, 4593380528125082431,
                                              16408922859458223821]


def test_seed_from_u64_is_four_splitmix_outputs():
    for seed in (0, 1, 9399220, M64):
        assert O.Xoshiro(seed).state() == O.splitmix64_stream(seed, 4)


# ---- (2) independent restatements -----------------------------------------------------------------
class PyXo:
    def __init__(self, s):
        self.s = list(s)

    def next(self):
        s = self.s
        res = (s[0] + s[3]) & M64
        t = (s[1] << 17) & M64
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = ((s[3] << 45) | (s[3] >> 19)) & M64
        return res


def py_uniform(rng, n):
    if n <= 0xFFFFFFFF:
        thresh = ((1 << 32) - n) % n
        while True:
            m = (rng.next() >> 32) * n
            if (m & 0xFFFFFFFF) >= thresh:
                return m >> 32
    thresh = ((1 << 64) - n) % n
    while True:
        m = rng.next() * n
        if (m & M64) >= thresh:
            return m >> 64


@pytest.mark.parametrize("n", [1, 2, 3, 10, 35059, 10_000_000, 0xFFFFFFFF, (1 << 32) + 7, (1 << 40) + 12345,
                               3 * (1 << 62)])
def test_uniform_usize_matches_python_lemire(n):
    a, b = O.Xoshiro(42), PyXo(O.splitmix64_stream(42, 4))
    for _ in range(200):
        assert a.uniform(n) == py_uniform(b, n)
    assert a.state() == b.s


def test_flip_and_f64_draws():
    a, b = O.Xoshiro(7), PyXo(O.splitmix64_stream(7, 4))
    for _ in range(100):
        assert a.flip() == b.next() >> 63
    for _ in range(100):
        assert a.random_f64() == (b.next() >> 11) * 2.0 ** -53


def py_sat_i32(v):
    if v != v:
        return 0
    return max(-2 ** 31, min(2 ** 31 - 1, int(v)))


def py_fpp(a, b):
    e = py_sat_i32(b)
    high = struct.unpack("<q", struct.pack("<d", a))[0] >> 32            # arithmetic: sign kept
    high = ((high + 2 ** 31) % 2 ** 32) - 2 ** 31
    diff = ((high - 1072632447 + 2 ** 31) % 2 ** 32) - 2 ** 31
    new_high = py_sat_i32((b - float(e)) * float(diff) + 1072632447.0)
    frac = struct.unpack("<d", struct.pack("<Q", (new_high % 2 ** 32) << 32))[0]
    base, r, ex = a, 1.0, e
    while ex != 0:
        if ex & 1:
            r *= base
        base *= base
        ex >>= 1
    return r * frac


def test_fast_precise_pow_matches_python_restatement():
    rng = np.random.default_rng(0)
    for a in list(rng.uniform(1e-9, 4.0, 200)) + [0.5, 1.0, 2.0, 1 / 3, 2 / 15931]:
        for b in [0.99, 0.01, 0.001, 0.999, 100.00000000000009, 1.001001001001001, 0.0, 1.0, 3.75]:
            got, want = O.fast_precise_pow(float(a), b), py_fpp(float(a), b)
            assert struct.pack("<d", got) == struct.pack("<d", want), (a, b)


def test_fast_precise_pow_known_values():
    # 1.0 has high word 0x3FF00000 = 1072693248; the magic literal is 1072632447 (sgd.rs:164)
    nh = int(0.99 * (1072693248 - 1072632447) + 1072632447.0)
    assert O.fast_precise_pow(1.0, 0.99) == struct.unpack("<d", struct.pack("<Q", nh << 32))[0]
    assert O.fast_precise_pow(2.0, 3.0) == 8.0 * O.fast_precise_pow(2.0, 0.0)
    assert abs(O.fast_precise_pow(0.5, 0.99) - 0.5 ** 0.99) < 0.02      # it IS approximate


def py_zipf(mn, mx, theta, zeta, z2, u):
    n = mx - mn + 1
    alpha = 1.0 / (1.0 - theta)
    eta = (1.0 - py_fpp(2.0 / float(n), 1.0 - theta)) / (1.0 - z2 / zeta)
    uz = u * zeta
    if uz < 1.0:
        return mn
    if uz < 1.0 + py_fpp(0.5, theta):
        return mn + 1
    res = float(mn) + float(n) * py_fpp(eta * u - eta + 1.0, alpha)
    if not (res > 0):
        r = 0
    elif res >= 2.0 ** 64:
        r = M64
    else:
        r = int(res)
    return min(r, mx)


def test_dirty_zipfian_matches_python_restatement():
    rng = np.random.default_rng(1)
    for theta in (0.99, 0.001):
        z2 = 1.0 + py_fpp(0.5, theta)
        for n in (1, 2, 3, 10, 100, 3100, 15931, 156249):
            p = O.params(theta=0.99, space=max(n, 1), space_max=100, space_quantization_step=100)
            zt = O.zetas(p)
            si = n if n <= 100 else 100 + (n - 100) // 100 + 1
            zeta = zt[min(si, len(zt) - 1)]
            for u in list(rng.uniform(0, 1, 50)) + [0.0, 1.0 - 2 ** -53]:
                assert O.dirty_zipfian(1, n, theta, zeta, z2, float(u)) == py_zipf(1, n, theta, zeta, z2, float(u))


def test_dirty_zipfian_second_fast_path_not_clamped():
    # sgd.rs:143-145 returns min+1 without clamping to max.  With the reference's own zeta table the
    # case max == min never gets there (zeta_1 < 1), so feed a zeta > 1 to show the inherited behaviour.
    z2 = 1.0 + O.fast_precise_pow(0.5, 0.99)
    got = {O.dirty_zipfian(1, 1, 0.99, 1.4, z2, k / 1000) for k in range(1000)}
    assert got == {1, 2}
    z1 = O.zetas(O.params(theta=0.99, space=1, space_max=100))[1]
    assert z1 < 1.0 and {O.dirty_zipfian(1, 1, 0.99, z1, z2, k / 1000) for k in range(1000)} == {1}


# ---- (3) values fixed by the reference's own files ------------------------------------------------
FIXTURE_TABLE = {   # SURVEY.md §4, derived from ygs.rs:50-93 / sgd.rs:733-762 on tests/data
    "simple.gfa": dict(nodes=15, steps=10, paths=1, Y=(10, 100.0, 50, 51), L=(100, 100.0, 10)),
    "lil.gfa": dict(nodes=15, steps=30, paths=3, Y=(30, 100.0, 50, 51), L=(300, 100.0, 10)),
    "DRB1-3123.gfa": dict(nodes=4955, steps=35059, paths=12, Y=(35059, 9610000.0, 15931, 260),
                          L=(350590, 9610000.0, 3100)),
}


@pytest.mark.parametrize("name", list(FIXTURE_TABLE))
def test_fixture_derived_parameters(name):
    want = FIXTURE_TABLE[name]
    g = load(name)
    assert (g.n_nodes, g.n_steps, g.n_paths) == (want["nodes"], want["steps"], want["paths"])
    y = P.YgsParams.from_graph(g, 0, 1).path_sgd
    assert (y.min_term_updates, y.eta_max, y.space) == want["Y"][:3]
    assert len(O.zetas(oracle_params(y))) == want["Y"][3]
    lp = P.LayoutSGDParams.from_graph(g, 2, 1)
    assert (lp.min_term_updates, lp.eta_max, lp.space) == want["L"]
    assert (lp.space_max, lp.iter_max, lp.space_quantization_step) == (1000, 30, 100)


def test_drb1_fixture_shape():
    g = load("DRB1-3123.gfa")
    assert int(g.step_is_rev.sum()) == 3096
    pos, plen = g.step_positions()
    assert int(plen.max()) == 15931 and int(g.path_step_counts().max()) == 3100
    assert int(g.node_len.sum()) == 21997


def test_path_index_simple_gfa():
    g = load("simple.gfa")
    pos, pth, rank, plen = O.path_index(oracle_graph(g))
    assert pos.tolist() == [0, 8, 9, 10, 13, 14, 33, 34, 38, 39]
    assert plen.tolist() == [50] and rank.tolist() == list(range(10)) and set(pth.tolist()) == {0}
    # host-side vectorised PathIndex agrees
    hpos, hlen = g.step_positions()
    assert hpos.tolist() == pos.tolist() and hlen.tolist() == plen.tolist()


def test_path_index_all_fixtures_host_vs_oracle():
    for name in FIXTURE_TABLE:
        g = load(name)
        pos, pth, rank, plen = O.path_index(oracle_graph(g))
        hpos, hlen = g.step_positions()
        assert np.array_equal(pos, hpos) and np.array_equal(plen, hlen)


def test_param_defaults():
    y = P.YgsParams()                        # ygs.rs:247-252
    assert (y.path_sgd.iter_max, y.path_sgd.theta, y.path_sgd.eps) == (100, 0.99, 0.01)
    d = P.PathSGDParams()                    # sgd.rs:214-234
    assert (d.min_term_updates, d.eta_max, d.space, d.space_max, d.space_quantization_step,
            d.cooling_start, d.seed, d.nthreads) == (100, 100.0, 100, 100, 100, 0.5, 9399220, 1)
    lp = P.LayoutSGDParams()                 # sgd.rs:709-729
    assert (lp.dimensions, lp.iter_max, lp.space_max) == (2, 30, 1000)


def test_schedule_closed_form():
    for eta_max, iters in [(100.0, 100), (9610000.0, 10), (24414062500.0, 200), (100.0, 30)]:
        p = O.params(eta_max=eta_max, iter_max=iters)
        e = O.schedule(p)
        assert len(e) == iters + 1
        em = 1.0 / (1.0 / eta_max)
        assert e[0] == em
        lam = math.log(em / 0.01) / (iters - 1.0)
        assert e[-1] == em * math.exp(-lam * iters)
        assert abs(e[-1] / (eta_max * (0.01 / eta_max) ** (iters / (iters - 1.0))) - 1) < 1e-9
        assert all(e[k] > e[k + 1] for k in range(iters))
    # iter_with_max_learning_rate shifts the peak (sgd.rs:633)
    p = O.params(eta_max=100.0, iter_max=10, iter_with_max_learning_rate=4)
    e = O.schedule(p)
    assert int(np.argmax(e)) == 4 and e[3] == e[5]


def test_schedule_iter_max_1_is_nan_like_reference():
    # lambda = ln(..)/0 = inf; t=0: exp(-inf*0) = NaN (inherited)
    e = O.schedule(O.params(eta_max=100.0, iter_max=1))
    assert math.isnan(e[0]) and e[1] == 0.0


def test_zeta_table_structure():
    p = O.params(theta=0.99, space=15931, space_max=100, space_quantization_step=100)
    z = O.zetas(p)
    assert len(z) == 100 + (15931 - 100) // 100 + 1 + 1 == 260
    assert z[0] == 0.0 and z[1] == O.fast_precise_pow(1.0, 0.99)
    acc = 0.0
    run = {}
    for i in range(1, 15932):
        acc += O.fast_precise_pow(1.0 / i, 0.99)
        run[i] = acc
    assert all(z[i] == run[i] for i in range(1, 101))
    assert z[101] == run[100]                          # i == space_max writes idx space_max+1 too
    assert all(z[101 + k] == run[100 + 100 * k] for k in range(0, 159))
    # space <= space_max: plain prefix table
    z2 = O.zetas(O.params(theta=0.99, space=50, space_max=100))
    assert len(z2) == 51 and all(z2[i] == run[i] for i in range(1, 51))


# ---- deterministic-mode invariants ------------------------------------------------------------------
def test_sgd_1d_counts_and_determinism():
    g = load("lil.gfa")
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    og, op = oracle_graph(g), oracle_params(p)
    xs = []
    for _ in range(2):
        x = O.init_positions(og)
        rc, st, _ = O.sgd_1d(og, op, x, n_streams=3)
        assert rc == 0 and st.term_updates == 101 * 30 and st.iterations == 101
        xs.append(x)
    assert np.array_equal(xs[0], xs[1])


def test_state_api_equals_one_shot():
    import ctypes as C
    g = load("DRB1-3123.gfa")
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = 4
    og, op = oracle_graph(g), oracle_params(p)
    x1 = O.init_positions(og)
    O.sgd_1d(og, op, x1, n_streams=5)
    L = O.lib()
    st = C.c_void_p()
    rc = L.gfo_state_create(og.ref, C.byref(op), None, None, C.c_uint64(0), C.c_uint64(5), C.c_uint64(0),
                            C.c_uint64(0), C.c_uint64(64), None, C.c_uint64(0), C.byref(st))
    assert rc == 0
    x2 = O.init_positions(og)
    for k in range(5):
        assert L.gfo_state_run_iteration(st, C.c_uint64(k), x2.ctypes.data_as(C.c_void_p)) == 0
    L.gfo_state_destroy(st)
    assert np.array_equal(x1, x2)


def test_nothing_to_do_cases():
    # empty graph / only single-step paths: reference returns an empty map (sgd.rs:242-244,258-261)
    g0 = O.Graph(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    rc, st, _ = O.sgd_1d(g0, O.params(), np.zeros(0), n_streams=1)
    assert rc == 1
    g1 = O.Graph(np.array([3, 4], np.uint32), np.array([0, 1], np.uint32), np.zeros(2, np.uint8),
                 np.array([0, 1, 2], np.uint64))
    rc, st, _ = O.sgd_1d(g1, O.params(), np.zeros(2), n_streams=1)
    assert rc == 1


def test_threaded_reference_like_mode_runs_and_counts():
    g = load("DRB1-3123.gfa")
    p = P.YgsParams.from_graph(g, 0, 2).path_sgd
    p.iter_max = 30
    og, op = oracle_graph(g), oracle_params(p)
    for flat in (0, 1):
        x = O.init_positions(og)
        s0 = O.stress_1d(og, x, 5000)
        rc, st = O.sgd_1d_threads(og, op, x, flat=flat)
        assert rc == 0 and st.iterations == 31 and st.term_updates >= 31 * p.min_term_updates
        assert np.isfinite(x).all() and O.stress_1d(og, x, 5000) < s0


def test_chain_graph_converges_to_chain_order():
    """P1: unique optimum — a block-shuffled linear chain sorts back into chain order (or its mirror)."""
    g = G.synth_chain(3000, 1)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd          # iter_max = 100, the CLI default
    og, op = oracle_graph(g), oracle_params(p)
    x = O.init_positions(og)
    O.sgd_1d(og, op, x, n_streams=4)
    ids = g.node_ids[np.argsort(x, kind="stable")].astype(np.int64)
    assert np.array_equal(ids, np.arange(1, 3001)) or np.array_equal(ids, np.arange(3000, 0, -1))


# ---- (4) committed golden file ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "oracle_golden.json")) as fh:
        return json.load(fh)


def _hex(a):
    return [format(int(v), "016x") for v in np.asarray(a, dtype=np.float64).view(np.uint64)]


def test_golden_scalars(golden):
    for a, b, h in golden["fast_precise_pow"]:
        assert _hex([O.fast_precise_pow(a, b)])[0] == h
    for mn, mx, th, zeta, z2, u, want in golden["dirty_zipfian"]:
        assert O.dirty_zipfian(mn, mx, th, zeta, z2, u) == want
    for seed, ent in golden["xoshiro"].items():
        r = O.Xoshiro(int(seed))
        assert r.state() == ent["state"] and [r.next_u64() for _ in range(8)] == ent["first8"]
    r = O.Xoshiro(9399220)
    u = golden["uniform_draws"]
    assert [r.uniform(10) for _ in range(16)] == u["n10"]
    assert [r.uniform(35059) for _ in range(16)] == u["n35059"]
    assert [r.flip() for _ in range(32)] == u["flips"]
    assert _hex([r.random_f64() for _ in range(8)]) == u["f64"]
    assert [r.uniform((1 << 33) + 5) for _ in range(8)] == u["n2pow33p5"]


def test_golden_tables(golden):
    for name, ent in golden["tables"].items():
        g = load(name)
        op = oracle_params(P.YgsParams.from_graph(g, 0, 1).path_sgd)
        e, z = O.schedule(op), O.zetas(op)
        assert _hex(e[:3]) == ent["etas_first3"] and _hex(e[-1:]) == ent["etas_last"]
        assert len(z) == ent["zetas_len"] and _hex(z[:6]) == ent["zetas_head"] and _hex(z[-4:]) == ent["zetas_tail"]


def test_golden_sgd_runs(golden):
    import hashlib
    for name, ent in golden["sgd_1d_single_stream"].items():
        g = load(name)
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = ent["iter_max"]
        og, op = oracle_graph(g), oracle_params(p)
        x = O.init_positions(og)
        rc, st, tr = O.sgd_1d(og, op, x, n_streams=1, trace_per_stream=8)
        assert (st.term_updates, st.attempts) == (ent["term_updates"], ent["attempts"])
        assert [[int(t["i"]), int(t["j"]), float(t["d_ij"])] for t in tr] == ent["trace8"]
        if "x" in ent:
            assert _hex(x) == ent["x"]
        else:
            assert _hex(x[:32]) == ent["x_head"] and hashlib.sha256(x.tobytes()).hexdigest() == ent["x_sha256"]
    ent = golden["sgd_nd_single_stream"]["DRB1-3123.gfa"]
    g = load("DRB1-3123.gfa")
    p = P.LayoutSGDParams.from_graph(g, 2, 1)
    p.iter_max, p.min_term_updates = ent["iter_max"], ent["min_term_updates"]
    c = gaussian_init(g, 2, 7)
    rc, st, tr = O.sgd_nd(oracle_graph(g), oracle_params(p), c, n_streams=1, trace_per_stream=8)
    assert (st.term_updates, st.attempts) == (ent["term_updates"], ent["attempts"])
    assert _hex(c[:32]) == ent["coords_head"] and hashlib.sha256(c.tobytes()).hexdigest() == ent["coords_sha256"]


# ---- rand_distr StandardNormal (ziggurat), restated: PARITY UNPINNED (the crate is not in the container) ------------
def test_ziggurat_tables_follow_the_published_construction():
    """256 layers of equal area V under exp(-x^2/2), x[1] = R, x[256] = 0; x[0] = V / f(R) (the base strip's virtual
    width).  3.910757959537090045 / 3.449278298560964462 / 0.000477467764586655 are the first table literals of
    rand_distr's ziggurat_tables.rs as remembered — not read from the crate here."""
    x, f = O.ziggurat_tables()
    R, V = 3.6541528853610088, 0.00492867323399
    assert x[1] == R and x[256] == 0.0 and f[256] == 1.0
    assert abs(x[0] - 3.910757959537090045) < 1e-14 and abs(x[2] - 3.449278298560964462) < 1e-14
    assert abs(f[0] - 0.000477467764586655) < 1e-17
    assert np.all(np.diff(x) < 0) and np.all(np.diff(f) > 0)
    assert np.allclose(f, np.exp(-x * x / 2.0), rtol=0, atol=1e-15)            # (numpy exp and libm exp differ by an ulp)
    areas = x[1:256] * (f[2:257] - f[1:256])                      # every strip above the base has area V
    assert np.abs(areas - V).max() < 1e-10


def test_standard_normal_statistics_and_stream_use():
    z = O.standard_normal(2024, 400_000)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1.0) < 0.01
    assert abs(((z - z.mean()) ** 4).mean() / z.var() ** 2 - 3.0) < 0.05
    from math import erf, sqrt
    for t in (1.0, 2.0, 3.0):
        want = 1.0 - erf(t / sqrt(2.0))
        assert abs((np.abs(z) > t).mean() - want) < 5.0 * np.sqrt(want / z.shape[0])
    # the common case consumes exactly one u64: i = low 8 bits, u from the top 52
    rng = O.Xoshiro(seed=99)
    bits = rng.next_u64()
    x, f = O.ziggurat_tables()
    i = bits & 0xFF
    u = np.frombuffer(np.uint64((bits >> 12) | (1024 << 52)).tobytes(), dtype=np.float64)[0] - 3.0
    if abs(u * x[i]) < x[i + 1]:
        assert O.standard_normal(99, 1)[0] == u * x[i]


@pytest.mark.parametrize("name", ["simple.gfa", "lil.gfa", "DRB1-3123.gfa"])
def test_layout_start_product_equals_oracle_and_follows_the_reference_draw_order(name):
    """sgd.rs:829-853: one generator; per node the + end's dims 1..D-1, then the - end's; dim 0 = prefix / prefix + len."""
    from gfasort_amd import hip, sgd as S
    g = load(name)
    og = oracle_graph(g)
    for D in (1, 2, 3):
        c = hip.init_layout(g, D, 9399220)
        assert np.array_equal(c.view(np.uint64), O.init_layout(og, D, 9399220).view(np.uint64))
        assert np.array_equal(S.default_layout_init(g, D, 9399220), c)
        c3 = c.reshape(g.n_nodes, 2, D)
        assert np.array_equal(c3[:, :, 0].reshape(-1), O.init_layout_dim0(og, 1))
        if D > 1:
            z = O.standard_normal(9399220, g.n_nodes * 2 * (D - 1)) * np.sqrt(2.0 * g.n_nodes)
            assert np.array_equal(c3[:, :, 1:].reshape(-1), z)       # node-major, + end first, dims ascending
