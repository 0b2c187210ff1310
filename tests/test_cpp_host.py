"""The C++ host mirror and its CLI (gfasort_amd/csrc/host): unit self-test on CPU; on the GPU box the
CLI runs `-p Y` / `-p L` end to end (GFA in -> GFA / TSV out) and is checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from util import O, G, P, ROOT, DATA, load, oracle_graph, oracle_params
from gfasort_amd import build as B
from gfasort_amd import hip


@pytest.fixture(scope="module")
def bins():
    B.build_host()
    return B.CLI, B.SELFTEST


def test_host_selftest(bins):
    out = subprocess.run([bins[1], DATA], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().endswith("ALL OK") and "ok fixture DRB1-3123.gfa" in out.stdout


def test_cli_rejects_out_of_scope_steps_and_bad_args(bins, tmp_path):
    o = str(tmp_path / "o.gfa")
    r = subprocess.run([bins[0], "-i", os.path.join(DATA, "simple.gfa"), "-o", o, "-p", "Yg"], capture_output=True, text=True)
    assert r.returncode == 1 and "not of this build" in r.stderr
    r = subprocess.run([bins[0], "-i", os.path.join(DATA, "simple.gfa"), "-o", o, "-p", "Q"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown pipeline character 'Q'" in r.stderr
    r = subprocess.run([bins[0], "-i", str(tmp_path / "missing.gfa"), "-o", o, "-p", "Y"], capture_output=True, text=True)
    assert r.returncode == 1 and "Error reading file" in r.stderr
    bad = tmp_path / "bad.gfa"
    bad.write_text("S\tx1\tA\n")
    r = subprocess.run([bins[0], "-i", str(bad), "-o", o, "-p", "Y"], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to parse node ID" in r.stderr


@pytest.mark.skipif(hip.lib().gfs_device_count() > 0, reason="a GPU is present")
def test_cli_fails_loudly_without_gpu(bins, tmp_path):
    o = tmp_path / "o.gfa"
    r = subprocess.run([bins[0], "-i", os.path.join(DATA, "simple.gfa"), "-o", str(o), "-p", "Y"], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr and not o.exists()


def _parse_out(path):
    g = G.load_gfa(path)
    with open(path) as fh:
        lines = fh.read().split("\n")
    return g, lines


@pytest.mark.gpu
@pytest.mark.parametrize("name,iters", [("simple.gfa", 100), ("lil.gfa", 100), ("DRB1-3123.gfa", 10)])
def test_cli_sort_replay_matches_oracle_order(bins, tmp_path, name, iters):
    """BASELINE configs[0] plumbing: `-p Y --iter-max N --streams 1` on the reference's fixtures — the
    written GFA equals apply_ordering(oracle order)."""
    src = os.path.join(DATA, name)
    o = str(tmp_path / "sorted.gfa")
    r = subprocess.run([bins[0], "-i", src, "-o", o, "-p", "Y", "--iter-max", str(iters), "--streams", "1", "-v", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    g = load(name)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = iters
    og = oracle_graph(g)
    x = O.init_positions(og)
    O.sgd_1d(og, oracle_params(p), x, n_streams=1)
    want = G.write_gfa_sorted(g, np.argsort(x, kind="stable"))
    with open(o) as fh:
        got = fh.read()
    wl, gl = want.split("\n"), got.split("\n")
    assert [l for l in gl if l[:1] in "HSP"] == [l for l in wl if l[:1] in "HSP"]
    assert sorted(l for l in gl if l[:1] == "L") == sorted(l for l in wl if l[:1] == "L")
    assert f"{(iters + 1) * p.min_term_updates} term updates" in r.stderr


@pytest.mark.gpu
def test_cli_sort_chain_full_width(bins, tmp_path):
    """P1 end to end: a block-shuffled chain comes out in chain order (or mirrored)."""
    g = G.synth_chain(20000, 1)
    src = tmp_path / "chain.gfa"
    src.write_text(G.synth_to_gfa_text(g))
    o = str(tmp_path / "chain.sorted.gfa")
    r = subprocess.run([bins[0], "-i", str(src), "-o", o, "-p", "Y", "-v", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    g2 = G.load_gfa(o)
    assert g2.node_ids.tolist() == list(range(1, 20001))
    lens_by_old_id = np.empty(20001, dtype=np.int64)
    lens_by_old_id[g.node_ids.astype(np.int64)] = g.node_len
    chain = lens_by_old_id[1:]
    assert np.array_equal(g2.node_len, chain) or np.array_equal(g2.node_len, chain[::-1])
    # the path now walks ids 1..N (or N..1)
    ids = g2.step_node_id.astype(np.int64)
    assert np.array_equal(ids, np.arange(1, 20001)) or np.array_equal(ids, np.arange(20000, 0, -1))


@pytest.mark.gpu
def test_cli_layout_tsv(bins, tmp_path):
    src = os.path.join(DATA, "DRB1-3123.gfa")
    o, tsv = str(tmp_path / "o.gfa"), str(tmp_path / "l.tsv")
    r = subprocess.run([bins[0], "-i", src, "-o", o, "-p", "L", "--dimensions", "2", "--layout-out", tsv, "-v", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "layout stress:" in r.stderr
    from gfasort_amd.layout import Layout
    with open(tsv) as fh:
        lay = Layout.read_tsv(fh)
    assert (lay.dimensions, lay.num_nodes) == (2, 4955) and np.isfinite(lay.coords).all()
    # `L` does not reorder the graph: the output GFA keeps the input ids and sequences
    g_in, g_out = load("DRB1-3123.gfa"), G.load_gfa(o)
    assert sorted(g_out.node_ids.tolist()) == sorted(g_in.node_ids.tolist()) and g_out.n_steps == g_in.n_steps
    stress = float(r.stderr.split("layout stress:")[1].split()[0])
    g = load("DRB1-3123.gfa")
    assert abs(stress - O.layout_stress(oracle_graph(g), 2, lay.coords, 10000)) < 1e-5


@pytest.mark.gpu
def test_cli_reference_sampler_flag(bins, tmp_path):
    """`--reference-sampler` (= --bundle 1) on a graph the auto policy would give to the team kernel: the engine line reports the
    sampler that ran; both write a sorted GFA of the same nodes."""
    g = G.synth_bubbles(20_000, 8, 4)
    src = tmp_path / "b.gfa"
    first = g.path_first_step.astype(int)
    with open(src, "w") as fh:
        fh.write("H\tVN:Z:1.0\n")
        fh.write("".join(f"S\t{i}\t{'A' * l}\n" for i, l in zip(g.node_ids.tolist(), g.node_len.tolist())))
        for pth, name in enumerate(g.path_names):
            fh.write(f"P\t{name}\t" + ",".join(f"{i}+" for i in g.step_node_id[first[pth]:first[pth + 1]].tolist()) + "\t*\n")
    seen = {}
    for extra, want in (([], "(bundle 64)"), (["--reference-sampler"], "(bundle 1)")):
        o = str(tmp_path / ("o%d.gfa" % len(extra)))
        r = subprocess.run([bins[0], "-i", str(src), "-o", o, "-p", "Y", "--iter-max", "30", "-v", "1"] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert want in r.stderr, r.stderr
        seen[want] = G.load_gfa(o)
    a, b = seen["(bundle 64)"], seen["(bundle 1)"]
    assert a.n_nodes == b.n_nodes == g.n_nodes and a.n_steps == b.n_steps == g.n_steps


# ---- the multi-device C ABI from C++: one host thread per rank, gfs_rank_run with a caller-supplied collective ---------
@pytest.mark.gpu
@pytest.mark.parametrize("args", [["2"], ["4", "60000", "24", "6000", "0", "2"], ["3", "40000", "12", "8000", "2", "1"],
                                  ["5", "30000", "3", "12000"], ["3", "40000", "12", "8000", "0", "1", "500"],
                                  ["2", "40000", "12", "8000", "2", "1", "300"]])
def test_multi_rank_selftest_threads_and_callback(args):
    """gfasort_amd/bin/multi_rank_selftest (csrc/host/multi_selftest.cpp): R host threads, each a gfs_rank on device 0,
    the whole schedule through gfs_rank_run with a host-staged all-reduce callback — the shape of a Rust host with one
    thread per GPU.  Replicas identical, quotas sum exactly, the chain sorts exactly (1D); with 5 ranks
    on 3 paths two ranks are idle and still take part in every collective; with unvisited nodes (7th argument) those end
    where they started on every replica (the final sum of "what I own" must give every slot an owner)."""
    B.build_host()
    out = subprocess.run([B.MULTI_SELFTEST] + args, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok ranks " + args[0])
