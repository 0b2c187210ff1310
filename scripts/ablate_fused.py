"""Ablation of the fused team kernel on C3 (full default run): what does the kernel gain when one memory stream is removed?
(wrong results by construction: GFS_F_DBG_NO_ATOMICS skips the adds, GFS_F_DBG_NO_XLOADS the position loads)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfasort_amd import graph as G, params as P, hip

g = G.synth_windows(1_000_000, 64, 156_250, 2)
p = P.YgsParams.from_graph(g, 0, 1).path_sgd
p.iter_max = 200
ctx = hip.Context(g)
for name, fl in (("full", 0), ("no_atomics", hip.F_DBG_NO_ATOMICS), ("no_xloads", hip.F_DBG_NO_XLOADS),
                 ("no_atomics+no_xloads", hip.F_DBG_NO_ATOMICS | hip.F_DBG_NO_XLOADS), ("one_partner", hip.F_ONE_PARTNER),
                 ("one_partner no_atomics", hip.F_ONE_PARTNER | hip.F_DBG_NO_ATOMICS), ("one_partner no_xloads", hip.F_ONE_PARTNER | hip.F_DBG_NO_XLOADS)):
    for rep in range(2):
        ctx.setup_1d(p, hip.make_config(flags=fl))
        ctx.init_positions()
        ctx.run()
        st = ctx.stats()
    print(f"{name:28s} streams {st.n_streams} {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:7.2f} G updates/s", flush=True)
ctx.close()
