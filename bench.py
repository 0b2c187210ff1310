#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-guided SGD hot path on MI355X.

Metric (BASELINE.json): SGD term-updates/sec for `-p Y` on the 1M-node synthetic GFA
(configs[2]: windows(N=1e6, P=64, W=156250, seed=2) => 10M path steps, --iter-max 200).

A "step" is ONE SGD iteration = one launch of the 1D kernel = min_term_updates (1e7) term
updates; step s runs iteration k = s mod (iter_max+1) of the eta/cooling schedule, so the
default --steps 201 is exactly one whole `-p Y --iter-max 200` run.  Graph, positions and RNG
streams are resident in HBM before the timed region starts.

    python bench.py --gpus 1 --steps 201 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N>1: BASELINE configs[4] — windows(N=1e7, P=1024, W=97656, seed=3), 1e8 steps, -p Y --iter-max 100 — with the paths
sharded over the ranks (gfs_rank below the C ABI, gfasort_amd/distributed.py RankDriver) and ONE RCCL all-reduce per
merge window of the slots two or more ranks can move.  The graph is fixed, the ranks split each iteration's 1e8 term
updates => "strong" scaling; rank 0 also times the same workload on its GPU alone (`single_gpu_same_workload`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_1D = 64          # 2 step records x16 B + 2 position reads x8 B + 2 writes x8 B
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
ATOMIC_UNIT_PEAK = 23.6e9   # 64-B f64 atomic requests/s the memory side executes, whole chip, measured (profiles/r01/ubench_scattered_ops.log)


def build_workload(world=1):
    """N = 1: BASELINE configs[2] (the configuration the metric is quoted on).  N > 1: configs[4]."""
    from gfasort_amd import graph as G, params as P
    if world > 1:
        g = G.synth_windows(10_000_000, 1024, 97_656, 3)
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = 100
        return g, p, ("windows(N=10000000,P=1024,W=97656,seed=3): 10M nodes / 1024 paths / 1e8 steps, -p Y --iter-max 100, "
                      "1e8 term updates per step")
    g = G.synth_windows(1_000_000, 64, 156_250, 2)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = 200
    return g, p, ("windows(N=1000000,P=64,W=156250,seed=2): 1M nodes / 64 paths / 10M steps, -p Y --iter-max 200, "
                  "1e7 term updates per step")


def host_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota
    (the GPU box exposes all host CPUs but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(g, p, seconds=12.0):
    """Reference-like CPU port (oracle threaded mode: Hogwild workers + 1 ms checker, hash-map
    node lookup, 8-byte step arrays) on all host cores, bounded sample of the same workload."""
    from oracle import oracle as O
    cores = host_cores()
    og = O.Graph(g.node_len, g.step_node, g.step_is_rev, g.path_first_step)
    kw = {k: getattr(p, k) for k in ["iter_max", "iter_with_max_learning_rate", "min_term_updates", "delta", "eps",
                                     "eta_max", "theta", "space", "space_max", "space_quantization_step",
                                     "cooling_start", "seed"]}
    op = O.params(nthreads=cores, **kw)
    x = O.init_positions(og)
    etas, zts = O.schedule(op), O.zetas(op)
    rc, st = O.sgd_1d_threads(og, op, x, flat=0, max_seconds=seconds, etas=etas, zts=zts)
    val = st.term_updates / st.seconds if st.seconds > 0 else 0.0
    # optimised flat variant (no hash map, 16-byte records), shorter sample
    x2 = O.init_positions(og)
    rc2, st2 = O.sgd_1d_threads(og, op, x2, flat=1, max_seconds=seconds / 2, etas=etas, zts=zts)
    val_flat = st2.term_updates / st2.seconds if st2.seconds > 0 else 0.0
    # one worker thread ("1 thread, and all host cores"), same port, shorter sample
    op1 = O.params(nthreads=1, **kw)
    x3 = O.init_positions(og)
    rc3, st3 = O.sgd_1d_threads(og, op1, x3, flat=0, max_seconds=seconds / 2, etas=etas, zts=zts)
    val_1t = st3.term_updates / st3.seconds if st3.seconds > 0 else 0.0
    return {"value": val, "unit": "term-updates/s", "cores": cores, "kind": "port",
            "sample": f"same 1M-node/10M-step graph, reference-like Hogwild port for {st.seconds:.1f} s "
                      f"({st.term_updates} updates, {st.iterations} iterations reached)",
            "flat_variant_value": val_flat,
            "one_thread": {"value": val_1t, "cores": 1,
                           "sample": f"same port, 1 worker thread for {st3.seconds:.1f} s ({st3.term_updates} updates)"}}


def quality_leg(g, p, device_index, args):
    """The whole `-p Y --iter-max 200` run of the timed configuration, from the reference's start, with the flags of the
    timed region: exact update count, adjacent inversions of the sort against the chain the graph is (P1: 0 expected),
    sampled stress before/after (the formula of sgd.rs:1196, gfasort_amd/quality.py), measure_layout_quality's RMSE."""
    from gfasort_amd import hip, quality as Q
    ctx = hip.Context(g, device=device_index)
    ctx.setup_1d(p, hip.make_config(n_streams=args.streams, flags=args.flags | hip.F_BUNDLE(args.bundle), block_size=args.block))
    ctx.init_positions()
    x0 = ctx.download()
    t0 = time.perf_counter()
    ctx.run()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    x = ctx.download()
    order = ctx.sort_order().astype(np.int64)
    ctx.close()
    lq = Q.layout_quality(g, order)
    return {"iterations": int(st.iterations), "term_updates": int(st.term_updates),
            "expected_term_updates": int((p.iter_max + 1) * p.min_term_updates), "launches": int(st.launches),
            "seconds": dt, "inversions_vs_chain_order": Q.inversions_vs_chain(g.node_ids[order].astype(np.int64)),
            "sampled_stress_before": Q.sampled_stress(g, x0, 0, 200000), "sampled_stress_after": Q.sampled_stress(g, x, 0, 200000),
            "layout_quality_rmse_bp": lq["rmse"], "layout_quality_mae_bp": lq["mae"]}


def bubbles_leg(device_index, args):
    """A graph that is NOT a chain: synth_bubbles(400000, 24, 6) — 525k nodes, 24 haplotypes, SNP bubbles and insertions,
    the default `-p Y` (iter_max 100).  Default flags against reference streams (GFS_F_BUNDLE(1)) at equal update counts:
    rate, sampled stress (2M pairs), the worst ratio of the relative error over the octaves of path distance, Kendall tau."""
    from gfasort_amd import graph as G, params as P, hip, quality as Q
    g = G.synth_bubbles(400_000, 24, 6)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    ctx = hip.Context(g, device=device_index)
    res = {}
    for name, flags in (("default_flags", args.flags | hip.F_BUNDLE(args.bundle)), ("reference_streams", hip.F_BUNDLE(1))):
        ctx.setup_1d(p, hip.make_config(n_streams=args.streams, flags=flags, block_size=args.block))
        ctx.init_positions()
        ctx.run()
        st = ctx.stats()
        x = ctx.download()
        _, rms, _ = Q.stress_by_scale(g, x, 0, 1_000_000)
        res[name] = {"value": st.term_updates / (st.kernel_ms * 1e-3), "unit": "term-updates/s", "sampling_bundle": int(st.bundle),
                     "run_trips": int(st.run_trips), "n_streams": int(st.n_streams), "term_updates": int(st.term_updates),
                     "sampled_stress_2M_pairs": Q.sampled_stress(g, x, 0, 2_000_000),
                     "roofline_frac": st.term_updates * ALGO_BYTES_1D / (st.kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "_rms": rms, "_order": ctx.sort_order().astype(np.int64)}
    ctx.close()
    a, b = res["default_flags"], res["reference_streams"]
    ra = Q.ranks_of(b.pop("_order"))
    rb = Q.oriented(ra, Q.ranks_of(a.pop("_order")))
    ratio = a.pop("_rms") / b.pop("_rms")
    res["workload"] = "synth_bubbles(400000,24,6): 525000 nodes / 24 paths / 9.75e6 steps, -p Y --iter-max 100"
    res["stress_ratio_default_over_reference_streams"] = a["sampled_stress_2M_pairs"] / b["sampled_stress_2M_pairs"]
    res["worst_error_ratio_over_octaves_of_path_distance"] = float(ratio.max())
    res["kendall_tau_of_the_two_sorts"] = Q.kendall_tau(ra, rb)
    return res


def reference_streams_leg(g, p, device_index, args, steps=60):
    """Same workload with GFS_F_BUNDLE(1): every lane is one reference worker stream
    (the sampler that is bit-identical to the reference's per-thread sampling)."""
    from gfasort_amd import hip
    eng = hip.Context(g, device=device_index)
    eng.setup_1d(p, hip.make_config(n_streams=args.streams, flags=args.flags | hip.F_BUNDLE(1), block_size=args.block))
    eng.init_positions()
    for k in range(3):
        eng.run_iteration(k)
    eng.synchronize()
    s0 = eng.stats()
    t0 = time.perf_counter()
    for k in range(3, 3 + steps):
        eng.run_iteration(k % (int(p.iter_max) + 1))
    eng.synchronize()
    dt = time.perf_counter() - t0
    s1 = eng.stats()
    upd = s1.term_updates - s0.term_updates
    kms = (s1.kernel_ms - s0.kernel_ms) / steps
    eng.close()
    return {"value": upd / dt, "unit": "term-updates/s", "steps": steps, "sampling_bundle": 1,
            "avg_launch_ms": kms, "roofline_frac": (upd / steps) * ALGO_BYTES_1D / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def layout_leg(g, device_index, args, dims=2):
    """BASELINE configs[3]: the same graph, -p L --dimensions 2 (31 iterations of 1e8 updates), resident in HBM.  Plus what
    the `bubbles` leg does for the sort: the default layout kernel against reference streams (GFS_F_BUNDLE(1)) on the 525k-node
    bubble graph at equal update counts from the same start — layout stress (the reference's formula, sgd.rs:1196, 2M pairs)
    and the worst ratio of the relative error over the octaves of path distance — and the wall clock of the CLI's `-p L`."""
    from gfasort_amd import hip, params as P, sgd as S, graph as G, quality as Q
    p = P.LayoutSGDParams.from_graph(g, dims, 1)
    ctx = hip.Context(g, device=device_index)
    ctx.setup_nd(p, hip.make_config(n_streams=args.streams, flags=args.flags | hip.F_BUNDLE(args.bundle), block_size=args.block))
    ctx.upload(S.default_layout_init(g, dims, p.seed).ravel())
    ctx.run_iteration(0)
    ctx.synchronize()
    s0 = ctx.stats()
    t0 = time.perf_counter()
    ctx.run_range(list(range(1, int(p.iter_max) + 1)))
    ctx.synchronize()
    dt = time.perf_counter() - t0
    s1 = ctx.stats()
    ctx.close()
    upd = s1.term_updates - s0.term_updates
    launches = max(int(s1.launches - s0.launches), 1)
    kms = (s1.kernel_ms - s0.kernel_ms) / launches
    algo = 40 + 32 * dims                                  # 2 records x 16 B + 2 ends x D x (8 read + 8 write) + ...
    its = int(s1.iterations - s0.iterations)
    out = {"value": upd / dt, "unit": "term-updates/s", "dimensions": dims, "steps": its, "launches": launches,
           "sampling_bundle": int(s1.bundle), "run_trips": int(s1.run_trips), "avg_launch_ms": kms,
           "kernel_ms_per_iteration": kms * launches / max(its, 1), "algorithmic_bytes_per_update": algo,
           "roofline_frac": (upd / launches) * algo / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    # the same launch against the memory-side atomic units (requests per update from the committed PMC pass of this kernel)
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as fh:
            tl = json.load(fh).get("layout_2d") or {}
    except (OSError, ValueError):
        tl = {}
    if dims == 2 and tl.get("atomic_requests_per_update"):
        req = float(tl["atomic_requests_per_update"])
        out.update(atomic_requests_per_update=req, atomic_unit_frac=req * (upd / launches) / (kms * 1e-3) / ATOMIC_UNIT_PEAK,
                   counter_bytes_per_update=tl.get("hbm_bytes_per_update"), counters_source=tl.get("source"))
    # quality of the default layout kernel against reference streams, 525k-node bubble graph
    gb = G.synth_bubbles(400_000, 24, 6)
    pb = P.LayoutSGDParams.from_graph(gb, dims, 1)
    c0 = S.default_layout_init(gb, dims, pb.seed)
    res = {}
    for name, flags in (("default_flags", args.flags | hip.F_BUNDLE(args.bundle)), ("reference_streams", hip.F_BUNDLE(1))):
        rc, c, st = hip.path_linear_sgd_layout_raw(gb, pb, c0, cfg=hip.make_config(n_streams=args.streams, flags=flags, block_size=args.block))
        _, rms, _ = Q.stress_by_scale(gb, c, dims, 1_000_000)
        cc = np.asarray(c).reshape(-1, 2, dims)
        e2e = np.abs(np.sqrt(((cc[:, 0, :] - cc[:, 1, :]) ** 2).sum(axis=1)) - gb.node_len)     # |distance between a node's ends - its length|
        res[name] = {"value": st.term_updates / (st.kernel_ms * 1e-3), "unit": "term-updates/s", "sampling_bundle": int(st.bundle),
                     "term_updates": int(st.term_updates), "launches": int(st.launches),
                     "layout_stress_2M_pairs": Q.sampled_stress(gb, c, dims, 2_000_000),
                     "node_end_to_end_error_bp_median_mean": [float(np.median(e2e)), float(np.mean(e2e))], "_rms": rms}
    ratio = res["default_flags"].pop("_rms") / res["reference_streams"].pop("_rms")
    out["bubbles_525k"] = dict(res, workload="synth_bubbles(400000,24,6), -p L --dimensions %d --layout-iter 30" % dims,
                               stress_ratio_default_over_reference_streams=res["default_flags"]["layout_stress_2M_pairs"] /
                               res["reference_streams"]["layout_stress_2M_pairs"],
                               worst_error_ratio_over_octaves_of_path_distance=float(ratio.max()))
    return out


def _write_gfa(g, path):
    with open(path, "w") as fh:
        fh.write("H\tVN:Z:1.0\n")
        fh.write("".join(f"S\t{i}\t{'A' * l}\n" for i, l in zip(g.node_ids.tolist(), g.node_len.tolist())))
        fh.write("".join(f"L\t{i}\t+\t{i + 1}\t+\t0M\n" for i in range(1, g.n_nodes)))
        first = g.path_first_step.astype(np.int64)
        for pth, name in enumerate(g.path_names):
            fh.write(f"P\t{name}\t" + ",".join(f"{i}+" for i in g.step_node_id[first[pth]:first[pth + 1]].tolist()) + "\t*\n")


def wall_clock_legs(g, pipelines=("Y", "L")):
    """The other half of BASELINE's metric: wall-clock of the whole run, GFA text in -> GFA text (and layout TSV) out, through
    the C++ CLI (gfasort_amd/bin/gfasort_hip): `-p Y --iter-max 200` (configs[2]) and `-p L --dimensions 2` (configs[3],
    src/bin/gfasort.rs:265-292).  The input file is written once."""
    import subprocess
    import tempfile
    from gfasort_amd import build as B
    if not os.path.exists(B.CLI):
        return {p: None for p in pipelines}
    d = tempfile.mkdtemp(prefix="gfs_bench_")
    src, dst, tsv = os.path.join(d, "c3.gfa"), os.path.join(d, "c3.out.gfa"), os.path.join(d, "c3.layout.tsv")
    _write_gfa(g, src)
    res = {}
    for pipeline in pipelines:
        cmd = [B.CLI, "-i", src, "-o", dst, "-p", pipeline, "-v", "1"] + \
              (["--iter-max", "200"] if pipeline == "Y" else ["--dimensions", "2", "--layout-out", tsv])
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True)
        dt = time.perf_counter() - t0
        phases = [ln for ln in r.stderr.split("\n") if ln.startswith("[gfasort] done")]
        engine = [ln for ln in r.stderr.split("\n") if ln.startswith("[gfasort_hip]")]
        out = {"seconds": dt, "returncode": r.returncode, "input_mb": os.path.getsize(src) / 1e6,
               "command": "gfasort_hip -i c3.gfa -o c3.out.gfa " + " ".join(cmd[5:]).replace(tsv, "c3.layout.tsv"),
               "phases": phases[0] if phases else "", "engine": engine[0] if engine else ""}
        if pipeline == "L":
            out["layout_stress_10k_pairs"] = next((float(ln.split("layout stress:")[1].split()[0]) for ln in r.stderr.split("\n")
                                                   if "layout stress:" in ln), None)
            out["layout_tsv_mb"] = os.path.getsize(tsv) / 1e6 if os.path.exists(tsv) else None
        res[pipeline] = out
    for f in (src, dst, tsv):
        try:
            os.remove(f)
        except OSError:
            pass
    return res


def run_leg(name, args):
    """An extra leg in a child process (see --leg)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", name, "--streams", str(args.streams), "--flags", str(args.flags),
           "--bundle", str(args.bundle), "--block", str(args.block)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    lines = [ln for ln in r.stdout.split("\n") if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"leg {name} failed (rc {r.returncode})", "stderr_tail": r.stderr[-400:]}
    return json.loads(lines[-1])


class _StdoutToStderr:
    """File-descriptor level: libraries that print banners on stdout (RCCL prints its version block when the
    first communicator is created) must not end up next to the ONE JSON line this script owes its caller."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=201)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--priming", action="store_true",
                    help="profiling runs only: after the warm-up steps, one untimed launch of the timed shape (both dispatches of the fused "
                         "kernel are then of one shape and its rocprofv3 --stats average is the timed launch's duration); disclosed in "
                         "config.untimed_priming_launch.  Default: the warm-up is exactly the --warmup steps")
    ap.add_argument("--leg", default="", help="internal: run ONE extra leg in this process and print its JSON")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="only the timed region: skip the quality / bubbles / reference_streams / layout_2d legs (PMC passes)")
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--bundle", type=int, default=0, help="sampling bundle: 0 = library auto policy, 1 = reference streams")
    ap.add_argument("--merge-every", type=int, default=1,
                    help="N>1: iterations per merge window (one all-reduce per window); 1 = an all-reduce at every "
                         "iteration, as BASELINE's north_star has it")
    ap.add_argument("--whole-vector", action="store_true", help="N>1: exchange the whole position vector, not only the shared slots")
    ap.add_argument("--payload-f64", action="store_true", help="N>1: f64 exchange buffer instead of f32")
    ap.add_argument("--no-layout-leg", action="store_true", help="N>1: skip the sharded `-p L` leg")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--no-fuse", action="store_true",
                    help="one kernel launch per iteration instead of one fused persistent launch per merge window")
    args = ap.parse_args()

    if args.leg:
        # One extra leg per child process: the parent's rocprofv3 --stats then holds the timed region's kernels only
        # (its fused kernel exactly once), and a leg cannot disturb the timed context.
        from gfasort_amd import hip
        if args.no_fuse:
            args.flags |= hip.F_NO_FUSE
        g, p, _ = build_workload(1)
        leg = {"quality": lambda: quality_leg(g, p, 0, args), "bubbles": lambda: bubbles_leg(0, args),
               "reference_streams": lambda: reference_streams_leg(g, p, 0, args), "layout_2d": lambda: layout_leg(g, 0, args)}[args.leg]
        print(json.dumps(leg()), flush=True)
        return

    import torch
    import torch.distributed as dist
    from gfasort_amd import hip
    from gfasort_amd.distributed import RankDriver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU fallback")
    # GFS_BENCH_SHARE_DEVICE=1 (rehearsal on a 1-GPU box): all ranks use cuda:0 and gloo, because
    # RCCL refuses two ranks on one device.  The driver's real runs never set it.
    share = os.environ.get("GFS_BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _StdoutToStderr():
            if share:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
            # first collective here: communicator creation (and its banner) happens under the redirection
            t = torch.zeros(1, device="cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()

    g, p, workload = build_workload(world)
    M = int(p.min_term_updates)
    if args.no_fuse:
        args.flags |= hip.F_NO_FUSE
    runner = RankDriver(g, p, rank, world, dims=0, device_index=local_rank, streams_per_rank=args.streams,
                        flags=args.flags | hip.F_BUNDLE(args.bundle), block_size=args.block,
                        dist=dist if world > 1 else None, merge_every=args.merge_every, whole_vector=args.whole_vector,
                        payload_f64=args.payload_f64)
    n_sched = int(p.iter_max) + 1

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # warmup (untimed), then restore the initial state so the timed run is a true run from k=0
    runner.set_positions(None)                                      # the reference's start, computed on the device
    # (one launch per warm-up step: the fused kernel's only dispatch is then the timed one, so its rocprofv3
    # --stats average is directly the figure reported below)
    for s in range(args.warmup):
        runner.run_iteration(s % n_sched)
    sync_all()
    # (--priming, profiling runs only: ONE untimed launch of exactly the timed shape.  The first dispatch of a kernel function
    # spends ~0.12 ms between the start event and the kernel's first wave — profiles/r02/launch_gap.log — which the HIP events
    # of a timed first dispatch include and the kernel trace does not: 5 % of the driver's 20-step launch, 0.6 % of the default
    # 201-step one.  The default run does not prime: its timed number is what --warmup W --steps K says.)
    if args.priming:
        runner.run_range([s % n_sched for s in range(args.steps)])
        sync_all()
    runner.reset_streams()
    runner.set_positions(None)
    st0 = runner.stats()
    sync_all()

    t0 = time.perf_counter()
    runner.run_range([s % n_sched for s in range(args.steps)])
    sync_all()
    elapsed = time.perf_counter() - t0

    st1 = runner.stats()
    local_updates = st1.term_updates - st0.term_updates
    local_kernel_ms = st1.kernel_ms - st0.kernel_ms
    multi = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([float(local_updates)], dtype=torch.float64, device="cuda")
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_updates = int(u.item())
        # a short instrumented pass after the timed region: where a window's time goes (HIP events on rank 0's stream)
        runner.profile = True
        runner.run_range([s % n_sched for s in range(min(args.steps, 24))])
        tm = runner.collect_timing()
        runner.profile = False
        sync_all()
        info = runner.info
        # what actually carried the collective: the library torch.distributed reports, the communicator's size, the device every
        # rank bound (gathered, so that the record itself shows N distinct GPUs — or the rehearsal's shared one)
        devs = [None] * world
        dist.all_gather_object(devs, {"rank": rank, "local_rank": local_rank, "cuda_device": int(torch.cuda.current_device()),
                                      "device_name": torch.cuda.get_device_name(torch.cuda.current_device()),
                                      "uuid": str(getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "uuid", ""))})
        multi = {"collective": {"torch_distributed_backend": str(dist.get_backend()), "world_size": int(dist.get_world_size()),
                                "is_rccl": str(dist.get_backend()) == "nccl" and bool(getattr(torch.version, "hip", None)),
                                "shared_device_rehearsal": bool(share), "ranks": devs},
                 "merge_every": args.merge_every, "windows_profiled": tm["windows"],
                 "compute_ms_per_window": tm["compute_ms"] / max(tm["windows"], 1),
                 "exchange_ms_per_window": tm["exchange_ms"] / max(tm["windows"], 1),
                 "exchange_bytes_per_window": tm["exchange_bytes_per_window"],
                 "shared_slots": int(info.shared_slots), "n_nodes": g.n_nodes,
                 "payload": "f64" if args.payload_f64 else "f32", "rank0_quota": int(info.quota),
                 "note": "compute = this rank's kernels of the window + packing its moves; exchange = the all-reduce of "
                         "[delta, touched] over the shared slots + applying it"}
        # second measurement, same run: merge windows of 4 iterations (one fused launch + one exchange per window)
        if args.merge_every == 1:
            from gfasort_amd.distributed import _Windows
            runner.windows = _Windows(p.iter_max, 4, True)
            runner.reset_streams()
            runner.set_positions(None)
            sync_all()
            u0 = runner.stats().term_updates
            t4 = time.perf_counter()
            runner.run_range([s % n_sched for s in range(args.steps)])
            sync_all()
            dt4 = torch.tensor([time.perf_counter() - t4], dtype=torch.float64, device="cuda")
            dist.all_reduce(dt4, op=dist.ReduceOp.MAX)
            u4 = torch.tensor([float(runner.stats().term_updates - u0)], dtype=torch.float64, device="cuda")
            dist.all_reduce(u4, op=dist.ReduceOp.SUM)
            runner.profile = True
            runner.timing.update({"compute_ms": 0.0, "exchange_ms": 0.0, "windows": 0})
            runner.run_range([s % n_sched for s in range(min(args.steps, 24))])
            tm4 = runner.collect_timing()
            runner.profile = False
            sync_all()
            multi["merge_every_4"] = {"value": float(u4.item()) / float(dt4.item()), "unit": "term-updates/s", "steps": args.steps,
                                      "compute_ms_per_window": tm4["compute_ms"] / max(tm4["windows"], 1),
                                      "exchange_ms_per_window": tm4["exchange_ms"] / max(tm4["windows"], 1),
                                      "windows_profiled": tm4["windows"]}
            runner.windows = _Windows(p.iter_max, args.merge_every, True)
        # second leg: the layout step (`-p L --dimensions 2`) of the same graph, sharded the same way: coordinates of both ends
        # of the shared slots are exchanged per iteration (2 planes x 2 dims); a few iterations are enough for a rate
        if not args.no_layout_leg:
            from gfasort_amd import params as PP, sgd as SS
            pl = PP.LayoutSGDParams.from_graph(g, 2, 1)
            lay = RankDriver(g, pl, rank, world, dims=2, device_index=local_rank, streams_per_rank=args.streams,
                             flags=args.flags | hip.F_BUNDLE(args.bundle), block_size=args.block, dist=dist,
                             merge_every=args.merge_every, whole_vector=args.whole_vector, payload_f64=args.payload_f64)
            lay.set_positions(SS.default_layout_init(g, 2, pl.seed).ravel())
            lay.run_iteration(0)
            sync_all()
            ul0 = lay.stats().term_updates
            n_l = min(args.steps, 8)
            tl = time.perf_counter()
            lay.run_range(list(range(1, 1 + n_l)))
            sync_all()
            dtl = torch.tensor([time.perf_counter() - tl], dtype=torch.float64, device="cuda")
            dist.all_reduce(dtl, op=dist.ReduceOp.MAX)
            ul = torch.tensor([float(lay.stats().term_updates - ul0)], dtype=torch.float64, device="cuda")
            dist.all_reduce(ul, op=dist.ReduceOp.SUM)
            multi["layout_2d"] = {"value": float(ul.item()) / float(dtl.item()), "unit": "term-updates/s", "steps": n_l,
                                  "term_updates_per_step": int(pl.min_term_updates), "dimensions": 2,
                                  "exchange_bytes_per_window": int(lay.info.exchange_count) * (8 if args.payload_f64 else 4),
                                  "roofline_frac_per_gpu": float(ul.item()) / float(dtl.item()) * 104 / 1e9 / HBM_PEAK_GBS / world}
            lay.close()
            sync_all()
        # the same workload on ONE GPU (rank 0's), so that the scaling of THIS workload can be read off this line
        if rank == 0:
            ctx1 = hip.Context(g, device=local_rank)
            ctx1.setup_1d(p, hip.make_config(n_streams=args.streams, flags=args.flags | hip.F_BUNDLE(args.bundle), block_size=args.block))
            ctx1.init_positions()
            ctx1.run_iteration(0)
            ctx1.synchronize()
            t1 = time.perf_counter()
            ks1 = [s % n_sched for s in range(min(args.steps, 20))]
            ctx1.run_range(ks1)
            ctx1.synchronize()
            dt1 = time.perf_counter() - t1
            multi["single_gpu_same_workload"] = {"value": len(ks1) * M / dt1, "unit": "term-updates/s", "steps": len(ks1)}
            ctx1.close()
        sync_all()
    else:
        total_updates = int(local_updates)

    if rank == 0:
        value = total_updates / elapsed
        launches = max(int(st1.launches - st0.launches), 1)
        avg_kernel_s = (local_kernel_ms / launches) * 1e-3
        upd_per_launch = local_updates / launches
        achieved = upd_per_launch * ALGO_BYTES_1D / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        # HBM traffic is a PMC measurement (separate rocprofv3 --pmc passes, MI355X_MICROARCH.md) and cannot be taken
        # inside this run: the figure of the committed passes over this same command is scaled to this launch and LABELLED
        traffic, traffic_source, per_update, atomic_req = None, None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if world == 1 and os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    tj = json.load(fh)
                per_update = tj.get("hbm_bytes_per_update")
                atomic_req = tj.get("atomic_requests_per_update")
                traffic = per_update * upd_per_launch if per_update else None
                traffic_source = ("NOT measured in this run: " + str(tj.get("source", "profiles/traffic_latest.json")) +
                                  f" ({per_update:.1f} B per update) scaled to this launch's updates")
            except Exception:
                traffic = None
        out = {
            "metric": "SGD term-updates/sec, -p Y, 1M-node synthetic GFA",
            "value": value, "unit": "term-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "term_updates_per_step": M, "n_streams_per_gpu": int(st1.n_streams),
                       "sampling_bundle": int(st1.bundle), "run_trips": int(st1.run_trips),
                       "sampler": ("library default: runs of 64 consecutive steps x run_trips trips per sampled step a, two independent "
                                   "partner draws per a (GFS_F_ONE_PARTNER: one), the waves of a fused launch drawing each iteration's "
                                   "updates from a shared pool — a departure from the reference's independent terms, see DESIGN.md 3 "
                                   "and the quality / bubbles legs") if int(st1.bundle) == 64 and not (args.flags & 0x30) else
                                  "as selected by --flags",
                       "untimed_priming_launch": None if not args.priming else
                       f"--priming: after the {args.warmup} warm-up steps, the {args.steps} steps of the timed region once, untimed, from the "
                       "same start (first-dispatch latency of the kernel function, ~0.12 ms, is not kernel time)",
                       "parallelism": f"paths sharded x{world} (consecutive blocks), one all-reduce of [delta,touched] "
                                      f"over the slots two or more ranks can move, every {args.merge_every} iteration(s), through "
                                      f"torch.distributed backend '{dist.get_backend()}' (see multi_gpu.collective)"
                       if world > 1 else "single GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": ("gfs::sgd1d_team_fused_kernel" if launches < args.steps else "gfs::sgd1d_team_kernel")
                                   if int(st1.bundle) != 1 else "gfs::sgd1d_kernel",
                         "launches": launches, "iterations_per_launch": args.steps / launches,
                         "term_updates_per_launch": upd_per_launch,
                         "avg_launch_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_update": ALGO_BYTES_1D,
                         # the same launch against two other yardsticks, both from the committed PMC passes (profiles/traffic_latest.json),
                         # NOT measured in this run: the HBM bytes the counters saw per update, and the requests to the memory-side
                         # atomic units — what actually binds this kernel — against the rate a micro-benchmark measured for them
                         "frac_counter_bytes": (per_update * upd_per_launch / avg_kernel_s / 1e9 / HBM_PEAK_GBS)
                         if per_update and avg_kernel_s > 0 else None,
                         "counter_bytes_per_update": per_update,
                         "atomic_unit_frac": (atomic_req * upd_per_launch / avg_kernel_s / ATOMIC_UNIT_PEAK)
                         if atomic_req and avg_kernel_s > 0 else None,
                         "atomic_requests_per_update": atomic_req, "atomic_unit_peak_requests_per_s": ATOMIC_UNIT_PEAK},
            "total_term_updates": total_updates,
        }
        if multi is not None:
            out["multi_gpu"] = multi
        if world == 1 and not args.no_extra_legs:
            legs = ["quality", "bubbles"] + (["reference_streams"] if int(st1.bundle) != 1 else []) + ["layout_2d"]
            for name in legs:
                out[name] = run_leg(name, args)
        if world == 1 and not args.no_cpu_baseline:
            wc = wall_clock_legs(g)
            out["wall_clock_pY"], out["wall_clock_pL"] = wc["Y"], wc["L"]
            out["cpu_baseline"] = cpu_baseline(g, p)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
