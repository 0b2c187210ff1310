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

N>1: paths are sharded over ranks, positions replicated, one RCCL all-reduce of the position
deltas per iteration (gfasort_amd/distributed.py) — total work is fixed => "strong" scaling.
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


def build_workload():
    from gfasort_amd import graph as G, params as P
    g = G.synth_windows(1_000_000, 64, 156_250, 2)
    p = P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = 200
    return g, p


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
    return {"value": val, "unit": "term-updates/s", "cores": cores, "kind": "port",
            "sample": f"same 1M-node/10M-step graph, reference-like Hogwild port for {st.seconds:.1f} s "
                      f"({st.term_updates} updates, {st.iterations} iterations reached)",
            "flat_variant_value": val_flat}


def reference_streams_leg(g, p, device_index, args, steps=60):
    """Same workload with GFS_F_BUNDLE(1): every lane is one reference worker stream
    (the sampler that is bit-identical to the reference's per-thread sampling)."""
    import torch
    from gfasort_amd import hip
    from gfasort_amd.distributed import HipEngine
    eng = HipEngine(g, p, 0, int(p.min_term_updates), 0, args.streams, device_index=device_index,
                    flags=args.flags | hip.F_BUNDLE(1), block_size=args.block)
    eng.set_positions(hip.init_positions(g))
    for k in range(3):
        eng.run_iteration(k)
    torch.cuda.synchronize()
    s0 = eng.stats()
    t0 = time.perf_counter()
    for k in range(3, 3 + steps):
        eng.run_iteration(k % (int(p.iter_max) + 1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = eng.stats()
    upd = s1.term_updates - s0.term_updates
    kms = (s1.kernel_ms - s0.kernel_ms) / steps
    eng.close()
    return {"value": upd / dt, "unit": "term-updates/s", "steps": steps, "sampling_bundle": 1,
            "avg_launch_ms": kms, "roofline_frac": (upd / steps) * ALGO_BYTES_1D / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def layout_leg(g, device_index, args, dims=2):
    """BASELINE configs[3]: the same graph, -p L --dimensions 2 (31 iterations of 1e8 updates), resident in HBM."""
    import torch
    from gfasort_amd import hip, params as P, sgd as S
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
    return {"value": upd / dt, "unit": "term-updates/s", "dimensions": dims, "steps": launches,
            "sampling_bundle": int(s1.bundle), "avg_launch_ms": kms, "algorithmic_bytes_per_update": algo,
            "roofline_frac": (upd / launches) * algo / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def wall_clock_leg(g):
    """The other half of BASELINE's metric: wall-clock of the whole `-p Y --iter-max 200` run,
    GFA text in -> sorted GFA text out, through the C++ CLI (gfasort_amd/bin/gfasort_hip)."""
    import subprocess
    import tempfile
    from gfasort_amd import build as B
    if not os.path.exists(B.CLI):
        return None
    d = tempfile.mkdtemp(prefix="gfs_bench_")
    src, dst = os.path.join(d, "c3.gfa"), os.path.join(d, "c3.sorted.gfa")
    with open(src, "w") as fh:
        fh.write("H\tVN:Z:1.0\n")
        fh.write("".join(f"S\t{i}\t{'A' * l}\n" for i, l in zip(g.node_ids.tolist(), g.node_len.tolist())))
        fh.write("".join(f"L\t{i}\t+\t{i + 1}\t+\t0M\n" for i in range(1, g.n_nodes)))
        first = g.path_first_step.astype(np.int64)
        for pth, name in enumerate(g.path_names):
            fh.write(f"P\t{name}\t" + ",".join(f"{i}+" for i in g.step_node_id[first[pth]:first[pth + 1]].tolist()) + "\t*\n")
    t0 = time.perf_counter()
    r = subprocess.run([B.CLI, "-i", src, "-o", dst, "-p", "Y", "--iter-max", "200", "-v", "1"], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    phases = [ln for ln in r.stderr.split("\n") if ln.startswith("[gfasort] done")]
    out = {"seconds": dt, "returncode": r.returncode, "input_mb": os.path.getsize(src) / 1e6,
           "command": "gfasort_hip -i c3.gfa -o c3.sorted.gfa -p Y --iter-max 200", "phases": phases[0] if phases else ""}
    for f in (src, dst):
        try:
            os.remove(f)
        except OSError:
            pass
    return out


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
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--bundle", type=int, default=0, help="sampling bundle: 0 = library auto policy, 1 = reference streams")
    ap.add_argument("--merge-every", type=int, default=8,
                    help="N>1: iterations between replica merges (all-reduce); quality at 8 ranks measured in "
                         "profiles/r01/virtual_cluster.log")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--no-fuse", action="store_true",
                    help="one kernel launch per iteration instead of one fused persistent launch per merge window")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gfasort_amd import hip
    from gfasort_amd.distributed import ShardedSGD, hip_engine_factory

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

    g, p = build_workload()
    M = int(p.min_term_updates)
    if args.no_fuse:
        args.flags |= hip.F_NO_FUSE
    runner = ShardedSGD(g, p, rank, world,
                        hip_engine_factory(device_index=local_rank, flags=args.flags | hip.F_BUNDLE(args.bundle),
                                           block_size=args.block),
                        dims=0, streams_per_rank=args.streams, dist=dist if world > 1 else None,
                        merge_every=args.merge_every)
    x0 = hip.init_positions(g)
    n_sched = int(p.iter_max) + 1

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # warmup (untimed), then restore the initial state so the timed run is a true run from k=0
    runner.set_positions(x0)
    # (one launch per warm-up step: the fused kernel's only dispatch is then the timed one, so its rocprofv3
    # --stats average is directly the figure reported below)
    for s in range(args.warmup):
        runner.run_iteration(s % n_sched)
    sync_all()
    runner.engine.reset_streams()
    runner.set_positions(x0)
    st0 = runner.engine.stats()
    sync_all()

    t0 = time.perf_counter()
    runner.run_range([s % n_sched for s in range(args.steps)])
    sync_all()
    elapsed = time.perf_counter() - t0

    st1 = runner.engine.stats()
    local_updates = st1.term_updates - st0.term_updates
    local_kernel_ms = st1.kernel_ms - st0.kernel_ms
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([float(local_updates)], dtype=torch.float64, device="cuda")
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_updates = int(u.item())
    else:
        total_updates = int(local_updates)

    if rank == 0:
        value = total_updates / elapsed
        launches = max(int(st1.launches - st0.launches), 1)
        avg_kernel_s = (local_kernel_ms / launches) * 1e-3
        upd_per_launch = local_updates / launches
        achieved = upd_per_launch * ALGO_BYTES_1D / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    per_update = json.load(fh).get("hbm_bytes_per_update")
                traffic = per_update * upd_per_launch if per_update else None
            except Exception:
                traffic = None
        out = {
            "metric": "SGD term-updates/sec, -p Y, 1M-node synthetic GFA",
            "value": value, "unit": "term-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "windows(N=1000000,P=64,W=156250,seed=2): 1M nodes / 64 paths / 10M steps, "
                                   "-p Y --iter-max 200, 1e7 term updates per step",
                       "term_updates_per_step": M, "n_streams_per_gpu": int(st1.n_streams),
                       "sampling_bundle": int(st1.bundle),
                       "parallelism": f"paths sharded x{world}, positions replicated, f32 [delta,touched] all-reduce "
                                      f"every {args.merge_every} iterations"
                       if world > 1 else "single GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("gfs::sgd1d_team_fused_kernel" if launches < args.steps else "gfs::sgd1d_team_kernel")
                                   if int(st1.bundle) != 1 else "gfs::sgd1d_kernel",
                         "launches": launches, "iterations_per_launch": args.steps / launches,
                         "term_updates_per_launch": upd_per_launch,
                         "avg_launch_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_update": ALGO_BYTES_1D},
            "total_term_updates": total_updates,
        }
        if world == 1 and int(st1.bundle) != 1:
            out["reference_streams"] = reference_streams_leg(g, p, local_rank, args)
        if world == 1:
            out["layout_2d"] = layout_leg(g, local_rank, args)
        if world == 1 and not args.no_cpu_baseline:
            out["wall_clock_pY"] = wall_clock_leg(g)
            out["cpu_baseline"] = cpu_baseline(g, p)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
