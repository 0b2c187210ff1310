"""Bundle width on MEDIUM graphs (25k-250k nodes): stress at equal update counts and speed, auto stream count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfasort_amd import graph as G
from bundle_quality import study

def main():
    study("bubbles 20k sites x 16 hap", G.synth_bubbles(20000, 16, 5), [100], [1, 8, 16, 32, 64], seeds=3)
    study("bubbles 60k sites x 16 hap", G.synth_bubbles(60000, 16, 7), [100], [1, 8, 16, 32, 64], seeds=2)
    study("bubbles 150k sites x 24 hap", G.synth_bubbles(150000, 24, 8), [100], [1, 16, 32, 64], seeds=2)
    study("windows 200k nodes x 16 paths", G.synth_windows(200_000, 16, 125_000, 7), [100], [1, 16, 32, 64], seeds=2)
    study("windows 50k nodes x 8 paths", G.synth_windows(50_000, 8, 25_000, 6), [100], [1, 8, 16, 32, 64], seeds=2)

if __name__ == "__main__":
    main()
