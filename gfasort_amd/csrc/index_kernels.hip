// index_kernels.hip — the one-off passes either side of the SGD loop, on the device (rocPRIM for
// the scan and the radix sort; the kernels around them are hand-written):
//   K4  initial positions (src/sgd.rs:286-294): exclusive prefix sum of node lengths
//   K3  PathIndex::from_graph (src/sgd.rs:34-71): step positions = per-path exclusive prefix sum
//       of node lengths over the steps, written straight into the 16-byte step records
//   K6  path_sgd_sort's sort (src/sgd.rs:665-671): positions -> rank order
#include <hip/hip_runtime.h>
#include <cstring>
#include <stdint.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace gfs {

// ---- K3 ------------------------------------------------------------------------------------------
// len(s) = sequence length of the step's node (0 for an absent node, sgd.rs:52-54); len(S) = 0.  Fed to the
// scan through a transform iterator, so the lengths are never materialised (8 B per step less scratch).
struct StepLen {
    const uint32_t *step_node, *node_len;
    uint64_t n_steps;
    __host__ __device__ uint64_t operator()(uint64_t s) const {
        if (s >= n_steps) return 0;
        const uint32_t n = step_node[s];
        return n == 0xFFFFFFFFu ? 0ull : (uint64_t)node_len[n];
    }
};

// Crowding statistics of the nodes (sgd_device.h crowd_shift): cnt[n] = steps on node n; rep[n] = the most visits
// to n within any 64 consecutive steps of one path (a step counts its node's earlier visits among the 63 steps
// before it, not looking across the start of its path).
__global__ void node_stats_kernel(const uint32_t *step_node, const uint64_t *path_first, uint32_t n_paths, uint64_t n_steps,
                                  uint32_t *cnt, uint32_t *rep) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_steps; s += stride) {
        const uint32_t n = step_node[s];
        if (n == 0xFFFFFFFFu) continue;
        atomicAdd(&cnt[n], 1u);
        uint32_t lo = 0, hi = n_paths;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (path_first[mid] <= s) lo = mid; else hi = mid; }
        const uint64_t begin = path_first[lo], from = s - begin > 63 ? s - 63 : begin;
        uint32_t c = 1;
        for (uint64_t t = from; t < s; ++t) c += step_node[t] == n;
        if (c > 1 && rep[n] < c) atomicMax(&rep[n], c);
    }
}
__device__ __forceinline__ uint32_t ceil_log2_u32(uint32_t v) { return v <= 1 ? 0u : 32u - (uint32_t)__clz((int)(v - 1)); }

// rec[s] = { internal node slot | NO_NODE, path | rev<<31, pos lo, pos hi (23 bits) | a<<23 | b<<29 } with pos = scan[s] - scan[first(path)]
__global__ void fill_records_kernel(const uint32_t *step_node, const uint8_t *step_is_rev, const uint32_t *perm,
                                    const uint64_t *scan, const uint64_t *path_first, uint32_t n_paths,
                                    const uint32_t *cnt, const uint32_t *rep, uint4 *rec, uint64_t n_steps) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_steps; s += stride) {
        // path of step s: last p with path_first[p] <= s (empty paths share a boundary: take the last)
        uint32_t lo = 0, hi = n_paths;                 // invariant: path_first[lo] <= s < path_first[hi]
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (path_first[mid] <= s) lo = mid; else hi = mid; }
        const uint64_t pos = scan[s] - scan[path_first[lo]];
        const uint32_t n = step_node[s];
        uint4 r;
        r.x = n == 0xFFFFFFFFu ? 0xFFFFFFFFu : perm[n];
        uint32_t crowd = 0;
        if (n != 0xFFFFFFFFu) {
            const uint32_t a = ceil_log2_u32(cnt[n]), b = ceil_log2_u32(rep[n]);
            crowd = ((a > 63u ? 63u : a) << 23) | ((b > 7u ? 7u : b) << 29);
        }
        r.y = lo | ((uint32_t)(step_is_rev[s] & 1) << 31);
        r.z = (uint32_t)pos; r.w = ((uint32_t)(pos >> 32) & 0x7FFFFFu) | crowd;
        rec[s] = r;
    }
}

__global__ void path_len_kernel(const uint64_t *scan, const uint64_t *path_first, uint64_t *path_len, uint32_t n_paths) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_paths) path_len[p] = scan[path_first[p + 1]] - scan[path_first[p]];
}

// All pointers are device pointers; tmp holds (n_steps+1) u64 (the scan).  Synchronous.
hipError_t build_path_index_device(const uint32_t *d_step_node, const uint8_t *d_step_is_rev, const uint32_t *d_node_len,
                                   const uint32_t *d_perm, const uint64_t *d_path_first, uint32_t n_paths,
                                   uint64_t n_steps, uint64_t n_nodes, uint64_t *d_tmp, uint4 *d_rec, uint64_t *d_path_len) {
    uint64_t *d_scan = d_tmp;
    const unsigned blocks = 2048;
    auto lens = rocprim::make_transform_iterator(rocprim::counting_iterator<uint64_t>(0), StepLen{d_step_node, d_node_len, n_steps});
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, lens, d_scan, (uint64_t)0, (size_t)(n_steps + 1),
                                           rocprim::plus<uint64_t>(), 0);
    if (e != hipSuccess) return e;
    void *d_scan_tmp = nullptr;
    if ((e = hipMalloc(&d_scan_tmp, tmp_bytes ? tmp_bytes : 8)) != hipSuccess) return e;
    e = rocprim::exclusive_scan(d_scan_tmp, tmp_bytes, lens, d_scan, (uint64_t)0, (size_t)(n_steps + 1),
                                rocprim::plus<uint64_t>(), 0);
    uint32_t *d_cnt = nullptr, *d_rep = nullptr;
    if (e == hipSuccess) e = hipMalloc(&d_cnt, (n_nodes ? n_nodes : 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&d_rep, (n_nodes ? n_nodes : 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, (n_nodes ? n_nodes : 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(d_rep, 0, (n_nodes ? n_nodes : 1) * sizeof(uint32_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(node_stats_kernel, dim3(4096), dim3(256), 0, 0, d_step_node, d_path_first, n_paths, n_steps, d_cnt, d_rep);
        hipLaunchKernelGGL(fill_records_kernel, dim3(blocks), dim3(256), 0, 0, d_step_node, d_step_is_rev, d_perm, d_scan,
                           d_path_first, n_paths, d_cnt, d_rep, d_rec, n_steps);
        hipLaunchKernelGGL(path_len_kernel, dim3((n_paths + 255) / 256), dim3(256), 0, 0, d_scan, d_path_first, d_path_len, n_paths);
        e = hipGetLastError();
    }
    hipError_t e2 = hipDeviceSynchronize();
    (void)hipFree(d_scan_tmp); (void)hipFree(d_cnt); (void)hipFree(d_rep);
    return e != hipSuccess ? e : e2;
}

// ---- internal node layout (first-visit path order with branches placed where they branch off) -----------------------
// Base rule: nodes in the order the paths first step on them, nodes no path visits last in index order (min-reduction over
// the steps + stable sort, so that a 4.5e9-step graph takes 0.3 s instead of 25 s on one host core).  The same pass
// validates step_node (dense index < n_nodes, or NO_NODE).
// Refinement: the first visits of a path come in RUNS of consecutive steps.  A run that starts its path is a ROOT run and
// keeps its place.  Any other run BRANCHES OFF the node its path stepped on just before it (its anchor) — an alternative
// allele of a bubble, first seen by the 7th haplotype, branches off the node before the bubble — and is placed right after
// the root-run node its chain of anchors leads to, instead of where the 7th haplotype happens to come in step order.
// sort key of node v = 2*first[v] for a root-run node, 2*first[root(v)] + 1 otherwise; ties (branches of one root node)
// in first-visit order.  Why: a run of 64 consecutive path steps then touches neighbouring position words on a bubble
// graph too, not 8 lines of main chain plus a scattered line per alternative allele (bubble graphs of 0.5M / 2M nodes:
// 63.9 -> 77.6 and 68.2 -> 79.7 G updates/s, profiles/r02/relayout_probe.log; chains and window graphs are unchanged —
// they have no branches).  gfs_shared_node_layout (multi.hip) is the same rule on the host.
__global__ void first_visit_kernel(const uint32_t *step_node, uint64_t n_steps, uint64_t n_nodes,
                                   unsigned long long *first, int *bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_steps; s += stride) {
        const uint32_t n = step_node[s];
        if (n == 0xFFFFFFFFu) continue;
        if (n >= n_nodes) { *bad = 1; continue; }
        if (first[n] > s) atomicMin(&first[n], (unsigned long long)s);      // the plain read only skips work
    }
}
__global__ void iota_kernel(uint32_t *v, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) v[k] = (uint32_t)k;
}
__global__ void scatter_rank_kernel(const uint32_t *sorted_node, uint32_t *perm, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) perm[sorted_node[r]] = (uint32_t)r;
}
__device__ __forceinline__ bool is_path_start(const uint64_t *path_first, uint32_t n_paths, uint64_t s) {
    uint32_t lo = 0, hi = n_paths;                                         // first path whose first step is >= s
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (path_first[mid] < s) lo = mid + 1; else hi = mid; }
    return lo < n_paths && path_first[lo] == s;
}
// up[v] = v for the first node of a run of first visits (and for unvisited nodes), else the node visited just before it
__global__ void run_up_kernel(const uint32_t *step_node, const unsigned long long *first, const uint64_t *path_first, uint32_t n_paths,
                              uint64_t n_nodes, uint32_t *up) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_nodes; v += stride) {
        const unsigned long long f = first[v];
        uint32_t u = (uint32_t)v;
        if (f != ~0ull && f > 0 && !is_path_start(path_first, n_paths, f)) {
            const uint32_t p = step_node[f - 1];
            if (p != 0xFFFFFFFFu && first[p] == f - 1) u = p;               // the previous step was a first visit too
        }
        up[v] = u;
    }
}
// pointer jumping: out[v] = in[in[v]]
__global__ void jump_kernel(const uint32_t *in, uint32_t *out, uint64_t n_nodes, int *changed) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_nodes; v += stride) {
        const uint32_t a = in[v], b = in[a];
        out[v] = b;
        if (b != a) *changed = 1;
    }
}
// anc[v] = v for a node of a root run (and unvisited nodes), else the node its run branches off; head[v] = first node of v's run
__global__ void anchor_kernel(const uint32_t *step_node, const unsigned long long *first, const uint64_t *path_first, uint32_t n_paths,
                              const uint32_t *head, uint64_t n_nodes, uint32_t *anc) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_nodes; v += stride) {
        uint32_t a = (uint32_t)v;
        if (first[v] != ~0ull) {
            const unsigned long long fh = first[head[v]];
            if (fh > 0 && !is_path_start(path_first, n_paths, fh)) {
                const uint32_t p = step_node[fh - 1];
                if (p != 0xFFFFFFFFu) a = p;
            }
        }
        anc[v] = a;
    }
}
// keys in first-visit order: key[r] for the node ids_by_first[r]
__global__ void branch_key_kernel(const unsigned long long *first, const uint32_t *root, const uint32_t *ids_by_first, uint64_t n_nodes,
                                  unsigned long long *key) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_nodes; r += stride) {
        const uint32_t v = ids_by_first[r];
        const unsigned long long f = first[v];
        const uint32_t t = root[v];
        key[r] = f == ~0ull ? ~0ull : (t == v ? 2ull * f : 2ull * first[t] + 1ull);
    }
}

// d_perm: n_nodes u32 (device).  d_path_first: [n_paths + 1] u64 (device).  *bad_out = 1 when a step names a node >= n_nodes.
// Synchronous.
hipError_t first_visit_layout_device(const uint32_t *d_step_node, uint64_t n_steps, uint64_t n_nodes, const uint64_t *d_path_first,
                                     uint32_t n_paths, uint32_t *d_perm, int *bad_out) {
    *bad_out = 0;
    if (n_nodes == 0) return hipSuccess;
    unsigned long long *d_first = nullptr, *d_first_sorted = nullptr;
    uint32_t *d_ids = nullptr, *d_ids_sorted = nullptr, *d_a = nullptr, *d_b = nullptr;
    int *d_bad = nullptr;
    void *d_tmp = nullptr;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t err) {
        (void)hipFree(d_first); (void)hipFree(d_first_sorted); (void)hipFree(d_ids); (void)hipFree(d_ids_sorted);
        (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_bad); (void)hipFree(d_tmp);
        return err;
    };
    if ((e = hipMalloc(&d_first, n_nodes * 8)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_first_sorted, n_nodes * 8)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_ids, n_nodes * 4)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_ids_sorted, n_nodes * 4)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_a, n_nodes * 4)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_b, n_nodes * 4)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_bad, 2 * sizeof(int))) != hipSuccess) return done(e);
    if ((e = hipMemset(d_first, 0xFF, n_nodes * 8)) != hipSuccess) return done(e);
    if ((e = hipMemset(d_bad, 0, 2 * sizeof(int))) != hipSuccess) return done(e);
    if (n_steps) hipLaunchKernelGGL(first_visit_kernel, dim3(4096), dim3(256), 0, 0, d_step_node, n_steps, n_nodes, d_first, d_bad);
    if ((e = hipMemcpy(bad_out, d_bad, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return done(e);
    if (*bad_out) return done(hipSuccess);                                 // (the kernels below index by step_node)
    hipLaunchKernelGGL(iota_kernel, dim3(1024), dim3(256), 0, 0, d_ids, n_nodes);
    size_t tmp_bytes = 0;
    e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t *)d_first, (uint64_t *)d_first_sorted, d_ids, d_ids_sorted,
                                  (size_t)n_nodes, 0, 64, 0);
    if (e != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8)) != hipSuccess) return done(e);
    e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, (uint64_t *)d_first, (uint64_t *)d_first_sorted, d_ids, d_ids_sorted,
                                  (size_t)n_nodes, 0, 64, 0);                 // stable: unvisited nodes stay in index order
    if (e != hipSuccess) return done(e);
    if (n_steps && d_path_first) {
        // branches: run heads by pointer jumping along the runs, anchors, roots by pointer jumping along the anchors
        int *d_changed = d_bad + 1;
        auto jump_to_fixpoint = [&](uint32_t *&cur, uint32_t *&other) -> hipError_t {
            for (int round = 0; round < 40; ++round) {
                hipError_t err = hipMemset(d_changed, 0, sizeof(int));
                if (err != hipSuccess) return err;
                hipLaunchKernelGGL(jump_kernel, dim3(2048), dim3(256), 0, 0, cur, other, n_nodes, d_changed);
                std::swap(cur, other);
                int changed = 0;
                if ((err = hipMemcpy(&changed, d_changed, sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) return err;
                if (!changed) break;
            }
            return hipSuccess;
        };
        hipLaunchKernelGGL(run_up_kernel, dim3(2048), dim3(256), 0, 0, d_step_node, d_first, d_path_first, n_paths, n_nodes, d_a);
        if ((e = jump_to_fixpoint(d_a, d_b)) != hipSuccess) return done(e);                          // d_a = run head of every node
        hipLaunchKernelGGL(anchor_kernel, dim3(2048), dim3(256), 0, 0, d_step_node, d_first, d_path_first, n_paths, d_a, n_nodes, d_b);
        std::swap(d_a, d_b);
        if ((e = jump_to_fixpoint(d_a, d_b)) != hipSuccess) return done(e);                          // d_a = root of every node
        hipLaunchKernelGGL(branch_key_kernel, dim3(2048), dim3(256), 0, 0, d_first, d_a, d_ids_sorted, n_nodes, d_first_sorted);
        // stable sort by key of the ids already in first-visit order: (key, first visit)
        e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, (uint64_t *)d_first_sorted, (uint64_t *)d_first, d_ids_sorted, d_ids,
                                      (size_t)n_nodes, 0, 64, 0);
        if (e != hipSuccess) return done(e);
        std::swap(d_ids, d_ids_sorted);
    }
    hipLaunchKernelGGL(scatter_rank_kernel, dim3(1024), dim3(256), 0, 0, d_ids_sorted, d_perm, n_nodes);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    return done(hipDeviceSynchronize());
}

// ---- ABI order <-> device order of positions / coordinates ------------------------------------------------
// ABI: x[k] (1D, D = 0) or coords[(k*2 + end)*D + dim] by dense index k; device: x[perm[k]] or the end x dimension planes
// coords[(end*D + dim)*N + perm[k]] (sgd_device.h coord_ptr).  to_device = 1: abi -> dev, else dev -> abi.
__global__ void reorder_positions_kernel(const double *src, double *dst, const uint32_t *perm, uint64_t N, uint32_t D, int to_device) {
    const uint64_t W = D ? 2ull * D : 1ull, total = N * W;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const uint64_t k = t / W, r = t - k * W;                  // t walks the ABI order; r = end*D + dim
        const uint64_t dev = r * N + perm[k];
        if (to_device) dst[dev] = src[t]; else dst[t] = src[dev];
    }
}
hipError_t reorder_positions_device(const double *d_src, double *d_dst, const uint32_t *d_perm, uint64_t N, uint32_t D,
                                    int to_device, hipStream_t st) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(reorder_positions_kernel, dim3(2048), dim3(256), 0, st, d_src, d_dst, d_perm, N, D, to_device);
    return hipGetLastError();
}

// ---- K4: initial 1D positions (sgd.rs:286-294): exclusive prefix sum of node lengths in node_order ---------
__global__ void widen_len_kernel(const uint32_t *node_len, uint64_t *out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) out[k] = node_len[k];
}
__global__ void scatter_prefix_kernel(const uint64_t *prefix, const uint32_t *perm, double *x, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        x[perm[k]] = (double)prefix[k];                          // the reference accumulates in usize and converts (sgd.rs:291)
}
// x[perm[k]] = sum of node_len[0..k).  Synchronous.
hipError_t init_positions_device(const uint32_t *d_node_len, const uint32_t *d_perm, double *d_x, uint64_t n) {
    if (n == 0) return hipSuccess;
    uint64_t *d_len = nullptr, *d_scan = nullptr; void *d_tmp = nullptr;
    hipError_t e;
    auto done = [&](hipError_t err) { (void)hipFree(d_len); (void)hipFree(d_scan); (void)hipFree(d_tmp); return err; };
    if ((e = hipMalloc(&d_len, n * 8)) != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_scan, n * 8)) != hipSuccess) return done(e);
    hipLaunchKernelGGL(widen_len_kernel, dim3(1024), dim3(256), 0, 0, d_node_len, d_len, n);
    size_t tmp_bytes = 0;
    e = rocprim::exclusive_scan(nullptr, tmp_bytes, d_len, d_scan, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), 0);
    if (e != hipSuccess) return done(e);
    if ((e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8)) != hipSuccess) return done(e);
    e = rocprim::exclusive_scan(d_tmp, tmp_bytes, d_len, d_scan, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), 0);
    if (e != hipSuccess) return done(e);
    hipLaunchKernelGGL(scatter_prefix_kernel, dim3(1024), dim3(256), 0, 0, d_scan, d_perm, d_x, n);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    return done(hipDeviceSynchronize());
}

// ---- K6 ------------------------------------------------------------------------------------------
// Order-preserving u64 image of an f64: -0.0 is folded onto +0.0 (partial_cmp calls them equal, so the
// tie must be broken by index, sgd.rs:666); NaNs (never produced by a finite run) sort after all numbers.
__device__ __forceinline__ uint64_t orderable(double v) {
    if (v != v) return 0xFFFFFFFFFFFFFFFFull;
    if (v == 0.0) v = 0.0;
    uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// keys in DENSE-index order (the device vector is in layout order), so that the stable sort breaks ties by dense index
__global__ void sort_keys_kernel(const double *x_layout, const uint32_t *perm, uint64_t *keys, uint32_t *vals, uint64_t n,
                                 uint64_t stride_doubles) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        keys[k] = orderable(x_layout[(uint64_t)perm[k] * stride_doubles]);
        vals[k] = (uint32_t)k;
    }
}

// d_order_out[r] = dense index of the node of rank r.  d_tmp: 2*n u64 keys + 2*n u32 values.  Synchronous.
hipError_t sort_order_device(const double *d_x_layout, const uint32_t *d_perm, uint64_t n, uint64_t stride_doubles,
                             void *d_tmp, uint32_t **d_order_out) {
    uint64_t *k_in = reinterpret_cast<uint64_t *>(d_tmp), *k_out = k_in + n;
    uint32_t *v_in = reinterpret_cast<uint32_t *>(k_out + n), *v_out = v_in + n;
    hipLaunchKernelGGL(sort_keys_kernel, dim3(1024), dim3(256), 0, 0, d_x_layout, d_perm, k_in, v_in, n, stride_doubles);
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0, 64, 0);
    if (e != hipSuccess) return e;
    void *d_sort_tmp = nullptr;
    if ((e = hipMalloc(&d_sort_tmp, tmp_bytes ? tmp_bytes : 8)) != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(d_sort_tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n, 0, 64, 0);   // stable
    hipError_t e2 = hipDeviceSynchronize();
    (void)hipFree(d_sort_tmp);
    *d_order_out = v_out;
    return e != hipSuccess ? e : e2;
}

// loads this translation unit's code object (HIP loads modules on first use); see gfs_warmup
hipError_t warm_module_index() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&path_len_kernel));
}

}  // namespace gfs
