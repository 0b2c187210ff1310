"""Wall-clock of `gfasort_hip -p L --dimensions 2 --layout-out` end to end on the C3 graph (BASELINE configs[3])."""
import sys, os, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfasort_amd import graph as G, build as B
from wallclock_cli import write_gfa_fast

def main():
    g = G.synth_windows(1_000_000, 64, 156_250, 2)
    d = tempfile.mkdtemp()
    src, dst, lay = os.path.join(d, "c3.gfa"), os.path.join(d, "c3.out.gfa"), os.path.join(d, "c3.lay.tsv")
    write_gfa_fast(g, src)
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([B.CLI, "-i", src, "-o", dst, "-p", "L", "--dimensions", "2", "--layout-out", lay, "-v", "1"],
                           capture_output=True, text=True)
        dt = time.time() - t0
        print("rc", r.returncode, f"wall {dt:.2f}s  layout file {os.path.getsize(lay) / 1e6:.1f} MB")
        print("\n".join(l for l in r.stderr.split("\n") if "gfasort_hip" in l or "done" in l or "stress" in l or " ms" in l))

if __name__ == "__main__":
    main()
