"""C4 (-p L --dimensions D) for a few iterations: the workload of the nD rocprofv3 passes.
usage: nd_pmc.py [D] [bundle] [K = run length in trips, 0 = the library's choice]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfasort_amd import graph as G, params as P, hip, sgd as S

dims = int(sys.argv[1]) if len(sys.argv) > 1 else 2
bundle = int(sys.argv[2]) if len(sys.argv) > 2 else 0
chain = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = G.synth_windows(1_000_000, 64, 156_250, 2)
p = P.LayoutSGDParams.from_graph(g, dims, 1)
p.iter_max = 7
c0 = S.default_layout_init(g, dims, p.seed)
rc, c, st = hip.path_linear_sgd_layout_raw(g, p, c0, cfg=hip.make_config(flags=hip.F_BUNDLE(bundle) | hip.F_CHAIN(chain)))
print(f"D={dims} bundle {st.bundle} K {st.run_trips}: {st.term_updates} updates, {st.launches} launches, "
      f"{st.term_updates / (st.kernel_ms * 1e-3) / 1e9:.2f} G upd/s", flush=True)
