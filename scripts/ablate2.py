"""Ablations + stream-count / block sweep of the bundled 1D kernel on C3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfasort_amd import graph as G, params as P, hip

def run(g, p, flags, T=0, block=0, n_iter=10, k0=3):
    ctx = hip.Context(g)
    ctx.setup_1d(p, hip.make_config(n_streams=T, flags=flags, block_size=block))
    ctx.upload(hip.init_positions(g))
    for k in range(3):
        ctx.run_iteration(k)
    ctx.synchronize()
    s0 = ctx.stats()
    for k in range(k0, k0 + n_iter):
        ctx.run_iteration(k)
    ctx.synchronize()
    s1 = ctx.stats()
    ctx.close()
    return (s1.term_updates - s0.term_updates) / ((s1.kernel_ms - s0.kernel_ms) * 1e-3) / 1e9, (s1.attempts - s0.attempts) / max(1, (s1.term_updates - s0.term_updates))

def main():
  g = G.synth_windows(1_000_000, 64, 156_250, 2)
  p = P.YgsParams.from_graph(g, 0, 1).path_sgd
  p.iter_max = 200
  for B in (64, 16):
      for name, fl in [("full", 0), ("no_atomics", 0x100), ("no_xloads", 0x200), ("neither", 0x300)]:
          r1, a1 = run(g, p, fl | hip.F_BUNDLE(B), k0=3)
          r2, a2 = run(g, p, fl | hip.F_BUNDLE(B), k0=150)
          print(f"B={B} {name:12s} noncool {r1:7.3f} G/s (att/upd {a1:.3f})  cooling {r2:7.3f} G/s (att/upd {a2:.3f})", flush=True)
  for T in (65536, 131072, 262144, 524288, 1048576, 2097152):
      for block in (256, 512, 1024):
          r1, _ = run(g, p, hip.F_BUNDLE(64), T=T, block=block)
          print(f"B=64 T={T:8d} block={block:4d}: {r1:7.3f} G/s", flush=True)

if __name__ == "__main__":
    main()
