"""Wall-clock of `gfasort_hip -p Y` end to end (GFA text in -> sorted GFA text out) on the C3 graph."""
import sys, os, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, build as B

def write_gfa_fast(g, path):
    with open(path, "w") as fh:
        fh.write("H\tVN:Z:1.0\n")
        ids = g.node_ids.tolist(); lens = g.node_len.tolist()
        fh.write("".join(f"S\t{i}\t{'A' * l}\n" for i, l in zip(ids, lens)))
        n = g.n_nodes
        fh.write("".join(f"L\t{i}\t+\t{i + 1}\t+\t0M\n" for i in range(1, n)))
        first = g.path_first_step.astype(np.int64)
        for p, name in enumerate(g.path_names):
            ids = g.step_node_id[first[p]:first[p + 1]].tolist()
            fh.write(f"P\t{name}\t" + ",".join(f"{i}+" for i in ids) + "\t*\n")

def main():
    g = G.synth_windows(1_000_000, 64, 156_250, 2)
    d = tempfile.mkdtemp()
    src, dst = os.path.join(d, "c3.gfa"), os.path.join(d, "c3.sorted.gfa")
    t0 = time.time(); write_gfa_fast(g, src); print(f"wrote {os.path.getsize(src)/1e6:.1f} MB GFA in {time.time()-t0:.1f}s", flush=True)
    for extra in ([], ["--bundle", "1"]):
        t0 = time.time()
        r = subprocess.run([B.CLI, "-i", src, "-o", dst, "-p", "Y", "--iter-max", "200", "-v", "1"] + extra, capture_output=True, text=True)
        dt = time.time() - t0
        print(" ".join(extra), "rc", r.returncode, f"wall {dt:.2f}s")
        print("\n".join(l for l in r.stderr.split("\n") if "gfasort_hip" in l or "done" in l or "ms" in l))
    g2 = G.load_gfa(dst) if os.path.getsize(dst) < 4e8 else None

if __name__ == "__main__":
    main()
