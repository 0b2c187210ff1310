// sgd_kernels_nd.hip — K2: reference-stream kernels of path_linear_sgd_layout, D = 1..8.
#include "sgd_kernel_common.h"

namespace gfs {

// ------------------------------------------------------------------------------------------
// K2: nD, D compile-time.  coords in planes [end][dim][slot] (sgd_device.h coord_ptr); the trace
// speaks the reference's index 2*node+end (sgd.rs:1099-1103).
// Node lengths come from the step records themselves: pos[s+1]-pos[s] inside a path,
// path_len - pos[s] for a path's last step (identical to graph.nodes[id].sequence.len(),
// 0 for an absent node — sgd.rs:1051-1058 — because PathIndex positions are the exclusive
// prefix sum of exactly those lengths, sgd.rs:43-54).
// ------------------------------------------------------------------------------------------
// Everything of one loop trip that does not depend on the coordinates: the pair sampler, the node lengths, the two end
// flips and the rejections of sgd.rs:990-1103.  Returns false where the reference `continue`s.
struct RefTermND { uint32_t ni, nj; uint32_t ends; int crowd; double term_dist; };     // ends: bit 0 = end of i, bit 1 = end of j

template <bool LDS_TABLES>
__device__ __forceinline__ bool ref_sample_nd(const KArgs &a, const uint4 *path_tab, const double *zeta_tab, Rng &rng,
                                              const uint64_t step_idx, const uint4 &ra, RefTermND &t) {
    uint4 rb; uint64_t sa, sb; uint32_t cnt, path;
    if (!sample_pair_from<LDS_TABLES>(a, path_tab, zeta_tab, rng, step_idx, ra, rb, sa, sb, cnt, path)) return false;
    const uint64_t first = path_first(path_tab[path]);
    const uint64_t last_step = first + cnt - 1u;
    const uint64_t plen = a.path_len[path];
    uint64_t pa = rec_pos_u64(ra), pb = rec_pos_u64(rb);
    uint64_t na, nb;                       // position of the following step / path end
    if (sa == last_step) na = plen; else { uint4 n = a.step_rec[sa + 1u]; na = rec_pos_u64(n); }
    if (sb == last_step) nb = plen; else { uint4 n = a.step_rec[sb + 1u]; nb = rec_pos_u64(n); }
    double pos_a = (double)pa, pos_b = (double)pb;                                     // sgd.rs:1047-1048
    const double len_i = (double)(na - pa), len_j = (double)(nb - pb);                // :1051-1058
    const bool rev_i = (ra.y >> 31) != 0, rev_j = (rb.y >> 31) != 0;                   // :1061,1070
    bool oa = rng.flip() == 1u;                                                        // :1062
    if (oa) { pos_a += len_i; oa = !rev_i; } else { oa = rev_i; }                      // :1063-1068
    bool ob = rng.flip() == 1u;                                                        // :1071
    if (ob) { pos_b += len_j; ob = !rev_j; } else { ob = rev_j; }                      // :1072-1077
    t.term_dist = fabs(pos_a - pos_b);                                                 // :1080
    if (t.term_dist == 0.0) return false;                                              // :1081
    t.crowd = crowd_shift<false>(a, ra, rb);
    t.ni = ra.x; t.nj = rb.x;
    t.ends = (oa ? 1u : 0u) | (ob ? 2u : 0u);
    return t.ni != 0xFFFFFFFFu && t.nj != 0xFFFFFFFFu;                                 // :1089-1096
}

// The worker loop for `quota` successful updates (sgd.rs:988-1156).  As in K1 (sgd_kernels_1d.hip ref_run_1d) the next trip's
// step a is drawn, and its record requested, before the current term's adds are issued.
template <int D, bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__device__ __forceinline__ void ref_run_nd(const KArgs &a, const uint4 *path_tab, const double *zeta_tab, Rng &rng,
                                           const uint32_t quota, const uint64_t max_att, const uint32_t tid,
                                           uint32_t &done, uint32_t &att, uint32_t &ntr) {
    uint32_t d = 0; uint64_t t = 0;
    uint64_t s_a = 0; uint4 r_a = make_uint4(0, 0, 0, 0); bool drawn = false;         // the next trip's step a, when drawn ahead
    while (d < quota && t < max_att) {
        ++t;
        if (!drawn) { s_a = sample_step(a, rng); r_a = a.step_rec[s_a]; }              // :990
        drawn = false;
        RefTermND cur;
        if (!ref_sample_nd<LDS_TABLES>(a, path_tab, zeta_tab, rng, s_a, r_a, cur)) continue;
        const bool oa = (cur.ends & 1u) != 0u, ob = (cur.ends & 2u) != 0u;
        const uint64_t idx_i = (uint64_t)cur.ni * 2u + (oa ? 1u : 0u);                 // :1099-1103
        const uint64_t idx_j = (uint64_t)cur.nj * 2u + (ob ? 1u : 0u);
        double *ci = coord_ptr<D>(a, cur.ni, oa), *cj = coord_ptr<D>(a, cur.nj, ob);
        const uint64_t cs = coord_step(a);
        const double mu = crowd_scale(fmin(a.it.eta * (1.0 / cur.term_dist), 1.0), cur.crowd);   // :1085-1086
        double deltas[D];
        double mag_sq = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {                                                  // :1108-1113
            deltas[k] = load_pos<ATOMIC_LOADS>(ci + k * cs) - load_pos<ATOMIC_LOADS>(cj + k * cs);
            mag_sq += deltas[k] * deltas[k];
        }
        if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }                       // :1116-1119
        const double mag = sqrt(mag_sq);                                               // :1121
        const double delta = mu * (mag - cur.term_dist) / 2.0;                         // :1125
        const double r = delta / mag;                                                  // :1142
        const bool same = idx_i == idx_j;   // reference stores c_i-r then c_j+r from values
                                            // loaded before either store: the 2nd wins (:1145-1148)
        if (d + 1u < quota && t < max_att) { s_a = sample_step(a, rng); r_a = a.step_rec[s_a]; drawn = true; }   // the next trip's :990
#pragma unroll
        for (int k = 0; k < D; ++k) {                                                  // :1143-1149
            const double r_d = r * deltas[k];
            if (!same) add_pos(ci + k * cs, -r_d);
            add_pos(cj + k * cs, r_d);
        }
        ++d;                                                                           // :1151
        if (TRACE) {
            if (ntr < a.trace_per_stream) {
                TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ntr;
                tt->i = (uint32_t)idx_i; tt->j = (uint32_t)idx_j; tt->d = cur.term_dist;
                ++ntr;
            }
        }
    }
    done += d;
    att += t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
}

template <int D, bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__global__ void sgdnd_kernel(const KArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint4 *path_tab; const double *zeta_tab;
    stage_tables<LDS_TABLES>(a, smem, path_tab, zeta_tab);

    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = tid < a.n_streams;
    uint32_t done = 0, att = 0;
    if (live) {
        Rng rng;
        const uint64_t T = a.n_streams;
        rng.s0 = a.rng[tid]; rng.s1 = a.rng[T + tid]; rng.s2 = a.rng[2 * T + tid]; rng.s3 = a.rng[3 * T + tid];
        const uint32_t quota = a.quota_base + (tid < a.quota_rem ? 1u : 0u);
        const uint64_t max_att64 = (uint64_t)a.attempt_factor * quota + 1024u;
        const uint64_t max_att = max_att64 > 0xFFFFFFFFull ? 0xFFFFFFFFull : max_att64;
        uint32_t ntr = TRACE ? a.trace_cnt[tid] : 0;
        ref_run_nd<D, LDS_TABLES, ATOMIC_LOADS, TRACE>(a, path_tab, zeta_tab, rng, quota, max_att, tid, done, att, ntr);
        a.rng[tid] = rng.s0; a.rng[T + tid] = rng.s1; a.rng[2 * T + tid] = rng.s2; a.rng[3 * T + tid] = rng.s3;
        if (TRACE) a.trace_cnt[tid] = ntr;
    }
    flush_counters(a, done, att);
}

// K2d: the same streams, a range of iterations in one persistent launch with work pools (sgd_kernel_common.h
// ref_pooled_walk; K1d in sgd_kernels_1d.hip is the 1D form).
template <int D, bool LDS_TABLES>
__global__ void sgdnd_fused_kernel(const KArgs a0, const IterConsts *its, const uint32_t n_iters, uint32_t *pool) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint4 *path_tab; const double *zeta_tab;
    stage_tables<LDS_TABLES>(a0, smem, path_tab, zeta_tab);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if ((tid & ~63u) >= a0.n_streams) return;                          // waves without a live lane
    const bool live = tid < a0.n_streams;
    const uint64_t T = a0.n_streams;
    KArgs a = a0;
    Rng rng = {0, 0, 0, 0};
    if (live) { rng.s0 = a.rng[tid]; rng.s1 = a.rng[T + tid]; rng.s2 = a.rng[2 * T + tid]; rng.s3 = a.rng[3 * T + tid]; }
    uint32_t done = 0, att = 0, ntr = 0;
    ref_pooled_walk(a, its, n_iters, pool, tid, [&](const uint32_t share, const uint64_t max_att) {
        ref_run_nd<D, LDS_TABLES, true, false>(a, path_tab, zeta_tab, rng, share, max_att, tid, done, att, ntr);
    });
    if (live) { a.rng[tid] = rng.s0; a.rng[T + tid] = rng.s1; a.rng[2 * T + tid] = rng.s2; a.rng[3 * T + tid] = rng.s3; }
    flush_counters(a, done, att);
}

template <int D, bool L, bool A>
static hipError_t launch_nd_t(const KArgs &a, bool trace, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (trace) hipLaunchKernelGGL((sgdnd_kernel<D, L, A, true>), grid, block, lds, st, a);
    else       hipLaunchKernelGGL((sgdnd_kernel<D, L, A, false>), grid, block, lds, st, a);
    return hipGetLastError();
}
hipError_t launch_nd_ref(int dims, const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                         dim3 grid, dim3 block, size_t lds, hipStream_t st) {
#define GFS_ND_CASE(D)                                                                                   \
    case D:                                                                                              \
        if (lds_tables) return atomic_loads ? launch_nd_t<D, true, true>(a, trace, grid, block, lds, st)  \
                                            : launch_nd_t<D, true, false>(a, trace, grid, block, lds, st); \
        return atomic_loads ? launch_nd_t<D, false, true>(a, trace, grid, block, 0, st)                   \
                            : launch_nd_t<D, false, false>(a, trace, grid, block, 0, st);
    switch (dims) {
        GFS_ND_CASE(1) GFS_ND_CASE(2) GFS_ND_CASE(3) GFS_ND_CASE(4)
        GFS_ND_CASE(5) GFS_ND_CASE(6) GFS_ND_CASE(7) GFS_ND_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef GFS_ND_CASE
}

// reference streams, fused (K2d); pool: zeroed counters, pool_bytes(n_iters) of them
hipError_t launch_nd_ref_fused(int dims, const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables, uint32_t *pool,
                               dim3 grid, dim3 block, size_t lds, hipStream_t st) {
#define GFS_NDF_CASE(D)                                                                                              \
    case D:                                                                                                          \
        if (lds_tables) hipLaunchKernelGGL((sgdnd_fused_kernel<D, true>), grid, block, lds, st, a, d_its, n_iters, pool);  \
        else            hipLaunchKernelGGL((sgdnd_fused_kernel<D, false>), grid, block, 0, st, a, d_its, n_iters, pool);   \
        return hipGetLastError();
    switch (dims) {
        GFS_NDF_CASE(1) GFS_NDF_CASE(2) GFS_NDF_CASE(3) GFS_NDF_CASE(4)
        GFS_NDF_CASE(5) GFS_NDF_CASE(6) GFS_NDF_CASE(7) GFS_NDF_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef GFS_NDF_CASE
}

// loads this translation unit's code object (HIP loads modules on first use); see gfs_warmup
hipError_t warm_module_nd() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&sgdnd_kernel<2, true, true, false>));
}

}  // namespace gfs
