"""CPU only: the oracle's team-mode mirror (1D, 2-D, 3-D; two partners, twin / fused trips on and off, a short chunk) under
AddressSanitizer + UBSan.
    make -C oracle asan && LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
        ASAN_OPTIONS=detect_leaks=0 python scripts/oracle_mirror_sanitizers.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle.oracle as OO
OO._LIB = os.path.join(ROOT, "oracle", "libgfs_oracle_asan.so")
from util import O, G, P, oracle_graph, oracle_params, gaussian_init
import numpy as np
from gfasort_amd.distributed import path_order_layout
for dims in (0, 2, 3):
    g = G.synth_windows(40_000, 8, 20_000, 12)
    p = P.LayoutSGDParams.from_graph(g, dims, 1) if dims else P.YgsParams.from_graph(g, 0, 1).path_sgd
    p.iter_max = 6
    p.min_term_updates = 60_000
    og, op = oracle_graph(g), oracle_params(p)
    x = gaussian_init(g, dims, 5) if dims else np.asarray(O.init_positions(og))
    for partners, twin, fused in ((2, True, True), (2, False, True), (1, True, False)):
        st = O.State(og, op, dims=dims, n_streams=128, bundle=64, node_slots=path_order_layout(g), chain=16 if dims else 64,
                     partners=partners, twin_trip=twin, fused_trip=fused, chunk=512 if partners == 1 else 0)
        xx = x.copy(); st.run(xx); s = st.stats()
        print(dims, partners, twin, fused, s.term_updates, s.attempts, float(np.abs(xx).sum()) > 0, flush=True)
print("mirror under ASan/UBSan: ok")
print([l.split()[-1] for l in open("/proc/self/maps") if "gfs_oracle" in l][:2])
