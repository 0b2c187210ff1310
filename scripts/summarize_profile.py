#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/profile.sh (gpurun_out/prof/<tag>/) into the small files kept
under profiles/: <prefix>_kernel_stats.csv (the --stats table), <prefix>_pmc_summary.csv (per-launch and
per-SGD-iteration means of every counter, per kernel) and profiles/traffic_latest.json (what bench.py
reports as roofline.traffic).

A fused launch covers many iterations, so counters are also normalised per iteration: the SGD kernels'
the fused kernel's single dispatch is bench.py's timed region (--steps iterations; bench.py warms up with
per-iteration launches), every dispatch of the other kernels is one iteration.

usage: summarize_profile.py gpurun_out/prof/<tag> profiles/r01/<prefix> --steps 20 --warmup 3 [--updates 10000000]
"""
import argparse
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void ", "").replace("gfs::", "")
    return n.split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst_prefix")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--updates", type=float, default=1e7, help="term updates per iteration")
    ap.add_argument("--no-traffic", action="store_true", help="do not rewrite profiles/traffic_latest.json")
    ap.add_argument("--priming", action="store_true", help="bench.py ran with --priming (two launches of --steps iterations)")
    args = ap.parse_args()

    newest = lambda files: max(files, key=os.path.getmtime)      # gpurun merges runs: keep the latest
    ks = glob.glob(os.path.join(args.src, "trace", "*", "*_kernel_stats.csv"))
    if ks:
        # bench.py runs its extra legs in child processes, which rocprofv3 traces into files of their own: the parent —
        # the timed region — is the process that started first (lowest pid)
        pid = lambda f: int(os.path.basename(f).split("_")[0])
        parent = min(ks, key=pid)
        with open(parent) as fh, open(args.dst_prefix + "_kernel_stats.csv", "w") as out:
            for row in fh:                                       # keep the SGD kernels; drop rocPRIM's kilobyte-long names
                if row.startswith('"Name"') or "gfs::" in row.split('",')[0]:
                    out.write(row)
    # every dispatch of the SGD kernels in the timed process, in order (the --stats table only has averages: bench.py's warm-up
    # steps are one-iteration ranges through the same gfs_ctx_run_range the timed region uses, i.e. dispatches of the same fused
    # kernel function, so its --stats average mixes W one-iteration launches with the timed launch of --steps iterations)
    kt = glob.glob(os.path.join(args.src, "trace", "*", "*_kernel_trace.csv"))
    if kt:
        pid = lambda f: int(os.path.basename(f).split("_")[0])
        parent = min(kt, key=pid)
        rows = []
        with open(parent) as fh:
            for r in csv.DictReader(fh):
                if "gfs::sgd" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        rows.sort()
        with open(args.dst_prefix + "_dispatches.csv", "w") as out:
            out.write("order,kernel,duration_ns,sgd_iterations\n")
            n_fused = sum(1 for r in rows if "fused" in r[1])
            seen = 0
            for k, (t0, name, dur) in enumerate(rows):
                its = 1
                if "fused" in name:
                    seen += 1
                    its = args.steps if (seen == n_fused or (args.priming and seen == n_fused - 1)) else 1
                out.write("%d,\"%s\",%d,%d\n" % (k, name, dur, its))
    bj = os.path.join(args.src, "bench_trace.json")
    if os.path.exists(bj):
        shutil.copy(bj, args.dst_prefix + "_bench_under_rocprof.json")

    rows = []
    per_iter = {}
    for d in sorted(glob.glob(os.path.join(args.src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not files:
            continue
        # (kernel, counter) -> list of (dispatch id, value); a counter may be reported once per dimension: sum
        acc = defaultdict(lambda: defaultdict(float))
        with open(newest(files)) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                if "sgd" not in k:
                    continue
                acc[(k, r["Counter_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, c), disp in sorted(acc.items()):
            vals = [disp[i] for i in sorted(disp)]
            fused = "fused" in k
            if fused:
                # bench.py: W warm-up ranges of one iteration, then (with --priming: an untimed and) the timed launch of --steps iterations
                big = 2 if args.priming else 1
                its = [1] * max(len(vals) - big, 0) + [args.steps] * min(big, len(vals))
            else:
                its = [1] * len(vals)
            mean = sum(vals) / len(vals)
            piter = sum(vals) / sum(its) if its else float("nan")
            rows.append((k, os.path.basename(d), c, len(vals), mean, min(vals), max(vals), piter))
            per_iter[(k, c)] = piter

    with open(args.dst_prefix + "_pmc_summary.csv", "w") as fh:
        fh.write("kernel,pass,counter,dispatches,mean_per_launch,min,max,mean_per_sgd_iteration\n")
        for r in rows:
            fh.write("\"%s\",%s,%s,%d,%.6g,%.6g,%.6g,%.6g\n" % r)

    # traffic of the dominant (team) kernel
    team = [k for (k, c) in per_iter if "team" in k and c == "FETCH_SIZE"]
    if team and not args.no_traffic:
        k = sorted(team, key=lambda s: "fused" not in s)[0]
        fetch_kb = per_iter[(k, "FETCH_SIZE")]
        write_kb = per_iter.get((k, "WRITE_SIZE"), float("nan"))
        hbm = (2.0 * fetch_kb + write_kb) * 1024.0
        out = {
            "kernel": k,
            "source": os.path.basename(args.dst_prefix) + "_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, "
                      "separate passes, bench.py --steps %d --warmup %d)" % (args.steps, args.warmup),
            "fetch_size_kb_per_iteration": fetch_kb,
            "write_size_kb_per_iteration": write_kb,
            "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B: confirmed on a streaming read and on "
                          "1-KB record runs, profiles/r01/fetch_calibration.txt); WRITE_SIZE as reported",
            "term_updates_per_iteration": args.updates,
            "hbm_bytes_per_update": hbm / args.updates,
            "algorithmic_bytes_per_update": 64,
        }
        # requests that reached the memory-side atomic units (what binds the team kernels), same passes
        atom = [per_iter[(kk, c)] for (kk, c) in per_iter if kk == k and c.startswith("TCC_EA0_ATOMIC")]
        if atom:
            out["atomic_requests_per_update"] = atom[0] / args.updates
        root = os.path.dirname(os.path.dirname(os.path.abspath(args.dst_prefix)))
        tpath = os.path.join(root, "traffic_latest.json")
        try:
            with open(tpath) as fh:
                prev = json.load(fh)
        except (OSError, ValueError):
            prev = {}
        if "sgdnd" in k:
            # a profile of the layout launches: kept beside the headline kernel's figures, which stay as they are
            out["algorithmic_bytes_per_update"] = 104
            out["source"] = (os.path.basename(args.dst_prefix) + "_pmc_summary.csv (scripts/profile_nd.sh: rocprofv3 --pmc passes, one per "
                             "counter group, over scripts/nd_pmc.py = BASELINE configs[3] `-p L --dimensions 2`, one fused launch of "
                             "%d iterations of %g updates)" % (args.steps, args.updates))
            prev["layout_2d"] = out
            out = prev
        elif "layout_2d" in prev:
            out["layout_2d"] = prev["layout_2d"]
        with open(tpath, "w") as fh:
            json.dump(out, fh, indent=1)
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
