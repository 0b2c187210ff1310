"""Launch-shape sweep: streams per launch for reference streams and bundled sampling."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfasort_amd import graph as G, params as P, hip
from ablate2 import run

def main():
    graphs = [("C3", G.synth_windows(1_000_000, 64, 156_250, 2), 200),
              ("bub400k", G.synth_bubbles(400000, 24, 6), 100)]
    for name, g, iters in graphs:
        p = P.YgsParams.from_graph(g, 0, 1).path_sgd
        p.iter_max = iters
        for B in (1, 16, 32, 64):
            best = (0, 0)
            for T in (65536, 98304, 131072, 163840, 196608, 262144, 393216, 524288):
                r, _ = run(g, p, hip.F_BUNDLE(B), T=T, block=256, k0=3)
                r2, _ = run(g, p, hip.F_BUNDLE(B), T=T, block=256, k0=iters * 3 // 4)
                print(f"{name} B={B:2d} T={T:7d}: noncool {r:7.3f} cooling {r2:7.3f} G/s", flush=True)
                best = max(best, (r + r2, T))
            print(f"{name} B={B} best T={best[1]}", flush=True)

if __name__ == "__main__":
    main()
