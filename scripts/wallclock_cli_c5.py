"""`gfasort_hip -p Y` end to end on the C5 graph (10M nodes / 1024 paths / 1e8 steps) as a GFA file, one GPU."""
import sys, os, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfasort_amd import graph as G, build as B
from wallclock_cli import write_gfa_fast

def main():
    g = G.synth_windows(10_000_000, 1024, 97_656, 3)
    d = tempfile.mkdtemp()
    src, dst = os.path.join(d, "c5.gfa"), os.path.join(d, "c5.sorted.gfa")
    t0 = time.time(); write_gfa_fast(g, src)
    print(f"wrote {os.path.getsize(src) / 1e9:.2f} GB GFA in {time.time() - t0:.0f}s", flush=True)
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([B.CLI, "-i", src, "-o", dst, "-p", "Y", "-v", "1"], capture_output=True, text=True)
        print("rc", r.returncode, f"wall {time.time() - t0:.2f}s  output {os.path.getsize(dst) / 1e9:.2f} GB")
        print("\n".join(l for l in r.stderr.split("\n") if "term updates" in l or "done" in l or "loaded" in l or " ms" in l))
    # the sorted file lists the nodes in chain order: S lines carry ids 1..N, path steps ascend or descend by 1
    with open(dst) as fh:
        for line in fh:
            if line.startswith("P\t"):
                steps = line.split("\t")[2].split(",")[:2000]
                ids = [int(s[:-1]) for s in steps]
                dd = {ids[k + 1] - ids[k] for k in range(len(ids) - 1)}
                print("first path, first 2000 steps: id differences", dd)
                break
    os.remove(src); os.remove(dst)

if __name__ == "__main__":
    main()
