"""Streams per node on medium graphs with the team kernel: stress at equal update counts and speed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfasort_amd import graph as G
from bundle_quality import study

def main():
    for name, g in (("bubbles 20k sites x 16 hap", G.synth_bubbles(20000, 16, 5)),
                    ("bubbles 60k sites x 16 hap", G.synth_bubbles(60000, 16, 7)),
                    ("windows 100k nodes x 32 paths", G.synth_windows(100_000, 32, 31_250, 2)),
                    ("bubbles 150k sites x 24 hap", G.synth_bubbles(150000, 24, 8))):
        for frac, label in ((4, "N/4"), (2, "N/2"), (1, "N"), (0.5, "2N")):
            T = int(g.n_nodes / frac) // 64 * 64
            study(f"{name}  T={label}", g, [100], [64], seeds=2, T=T)

if __name__ == "__main__":
    main()
