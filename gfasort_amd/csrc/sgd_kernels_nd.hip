// sgd_kernels_nd.hip — K2: reference-stream kernels of path_linear_sgd_layout, D = 1..8.
#include "sgd_kernel_common.h"

namespace gfs {

// ------------------------------------------------------------------------------------------
// K2: nD, D compile-time.  coords in end planes [end][slot][dim] (sgd_device.h coord_ptr); the trace
// speaks the reference's index 2*node+end (sgd.rs:1099-1103).
// Node lengths come from the step records themselves: pos[s+1]-pos[s] inside a path,
// path_len - pos[s] for a path's last step (identical to graph.nodes[id].sequence.len(),
// 0 for an absent node — sgd.rs:1051-1058 — because PathIndex positions are the exclusive
// prefix sum of exactly those lengths, sgd.rs:43-54).
// ------------------------------------------------------------------------------------------
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
        const uint32_t max_att = max_att64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)max_att64;
        uint32_t ntr = TRACE ? a.trace_cnt[tid] : 0;
        while (done < quota && att < max_att) {
            ++att;
            uint4 ra, rb; uint64_t sa, sb; uint32_t cnt, path;
            if (!sample_pair<LDS_TABLES>(a, path_tab, zeta_tab, rng, ra, rb, sa, sb, cnt, path)) continue;
            const uint64_t first = path_first(path_tab[path]);
            const uint64_t last_step = first + cnt - 1u;
            const uint64_t plen = a.path_len[path];
            uint64_t pa = rec_pos_u64(ra), pb = rec_pos_u64(rb);
            uint64_t na, nb;                       // position of the following step / path end
            if (sa == last_step) na = plen; else { uint4 n = a.step_rec[sa + 1u]; na = rec_pos_u64(n); }
            if (sb == last_step) nb = plen; else { uint4 n = a.step_rec[sb + 1u]; nb = rec_pos_u64(n); }
            double pos_a = (double)pa, pos_b = (double)pb;                             // sgd.rs:1047-1048
            const double len_i = (double)(na - pa), len_j = (double)(nb - pb);        // :1051-1058
            const bool rev_i = (ra.y >> 31) != 0, rev_j = (rb.y >> 31) != 0;           // :1061,1070
            bool oa = rng.flip() == 1u;                                                // :1062
            if (oa) { pos_a += len_i; oa = !rev_i; } else { oa = rev_i; }              // :1063-1068
            bool ob = rng.flip() == 1u;                                                // :1071
            if (ob) { pos_b += len_j; ob = !rev_j; } else { ob = rev_j; }              // :1072-1077
            double term_dist = fabs(pos_a - pos_b);                                    // :1080
            if (term_dist == 0.0) continue;                                            // :1081
            double mu = crowd_scale(fmin(a.it.eta * (1.0 / term_dist), 1.0), crowd_shift<false>(a, ra, rb));                       // :1085-1086
            if (ra.x == 0xFFFFFFFFu || rb.x == 0xFFFFFFFFu) continue;                  // :1089-1096
            const uint64_t idx_i = (uint64_t)ra.x * 2u + (oa ? 1u : 0u);               // :1099-1103
            const uint64_t idx_j = (uint64_t)rb.x * 2u + (ob ? 1u : 0u);
            double *ci = coord_ptr<D>(a, ra.x, oa), *cj = coord_ptr<D>(a, rb.x, ob);
            double deltas[D];
            double mag_sq = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) {                                              // :1108-1113
                deltas[d] = load_pos<ATOMIC_LOADS>(ci + d) - load_pos<ATOMIC_LOADS>(cj + d);
                mag_sq += deltas[d] * deltas[d];
            }
            if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }                   // :1116-1119
            double mag = sqrt(mag_sq);                                                 // :1121
            double delta = mu * (mag - term_dist) / 2.0;                               // :1125
            double r = delta / mag;                                                    // :1142
            const bool same = idx_i == idx_j;   // reference stores c_i-r then c_j+r from values
                                                // loaded before either store: the 2nd wins (:1145-1148)
#pragma unroll
            for (int d = 0; d < D; ++d) {                                              // :1143-1149
                double r_d = r * deltas[d];
                if (!same) add_pos(ci + d, -r_d);
                add_pos(cj + d, r_d);
            }
            ++done;                                                                    // :1151
            if (TRACE) {
                if (ntr < a.trace_per_stream) {
                    TraceTerm *t = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ntr;
                    t->i = (uint32_t)idx_i; t->j = (uint32_t)idx_j; t->d = term_dist;
                    ++ntr;
                }
            }
        }
        a.rng[tid] = rng.s0; a.rng[T + tid] = rng.s1; a.rng[2 * T + tid] = rng.s2; a.rng[3 * T + tid] = rng.s3;
        if (TRACE) a.trace_cnt[tid] = ntr;
    }
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

// loads this translation unit's code object (HIP loads modules on first use); see gfs_warmup
hipError_t warm_module_nd() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&sgdnd_kernel<2, true, true, false>));
}

}  // namespace gfs
