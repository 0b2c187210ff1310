"""> 2^32 path steps on ONE MI355X: 2^28 nodes, 17 paths that each walk all nodes => 4.56e9 steps,
73 GB of 16-byte step records + 97 GB of build temporaries in HBM.  Two SGD iterations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gfasort_amd import graph as G, params as P, hip

def log(*a):
    print(f"[{time.time() - T0:7.1f}s]", *a, flush=True)

T0 = time.time()
N, NP = 1 << 28, 17
perm = (np.arange(N, dtype=np.uint64) * np.uint64(0x9E3779B1) % np.uint64(N)).astype(np.uint32)   # odd multiplier: a bijection
log("perm done")
step_node = np.tile(perm, NP)
log("step_node", step_node.shape, step_node.nbytes / 1e9, "GB")
g = G.FlatGraph(node_len=(1 + (np.arange(N, dtype=np.uint32) * np.uint32(2654435761) >> np.uint32(28))).astype(np.uint32),
                step_node=step_node, step_is_rev=np.zeros(step_node.shape[0], dtype=np.uint8),
                path_first_step=(np.arange(NP + 1, dtype=np.uint64) * np.uint64(N)),
                node_ids=np.arange(1, N + 1, dtype=np.uint64), path_names=[f"p{k}" for k in range(NP)])
log("graph assembled: steps", g.n_steps, "> 2^32:", g.n_steps > 2 ** 32)
p = P.PathSGDParams(iter_max=1, min_term_updates=g.n_steps, eta_max=float(N) ** 2, space=int(g.node_len.astype(np.uint64).sum()),
                    space_max=100, space_quantization_step=100)
log("params", p.min_term_updates, p.space)
ctx = hip.Context(g)
log("context created (PathIndex built on the device)")
ctx.setup_1d(p)
log("setup done")
x0 = hip.init_positions(g)
ctx.upload(x0)
log("positions uploaded")
for k in range(2):
    ctx.run_iteration(k)
    ctx.synchronize()
    st = ctx.stats()
    log(f"iteration {k}: updates {st.term_updates} streams {st.n_streams} bundle {st.bundle} kernel_ms {st.kernel_ms:.1f}")
x = ctx.download()
st = ctx.stats()
log("finite:", bool(np.isfinite(x).all()), "moved nodes:", int((x != x0).sum()), f"rate {st.term_updates / (st.kernel_ms * 1e-3) / 1e9:.2f} G updates/s")
assert st.term_updates == 2 * g.n_steps and np.isfinite(x).all()
ctx.close()
log("OK")
