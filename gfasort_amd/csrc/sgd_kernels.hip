// sgd_kernels.hip — the SGD batch kernels (gfx950 / CDNA4, wave64).
//
// K1  sgd1d_kernel : one launch = one SGD iteration of path_linear_sgd      (src/sgd.rs:442-584)
// K2  sgdnd_kernel : one launch = one iteration of path_linear_sgd_layout   (src/sgd.rs:988-1156)
//
// Execution model: one lane = one Xoshiro256+ stream = one reference worker thread
// (seed + stream id, sgd.rs:431-432).  A launch replaces the reference's checker thread
// (sgd.rs:366-407): eta / theta / cooling are launch constants and every stream performs
// exactly its quota of successful term updates, so the number of updates per iteration is
// min_term_updates, not wall-clock dependent.
//
// Memory: this is an HBM/fabric-bound gather/scatter, no MFMA.  Per update the kernel touches
// two 16-B step records (random), two (1D) position words read with agent-scope relaxed
// atomic loads, and two no-return f64 atomic adds (global_atomic_add_f64: gfx950 has the
// native instruction, so no CAS loop).  The zeta table and the per-path records are staged
// once per workgroup into LDS.  RNG state lives in registers for the whole launch and is
// loaded/stored coalesced (SoA) at entry/exit.
#include "sgd_device.h"

namespace gfs {

struct TraceTerm { uint32_t i, j; double d; };

template <bool ATOMIC_LOADS>
__device__ __forceinline__ double load_pos(const double *p) {
    if (ATOMIC_LOADS)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
__device__ __forceinline__ void add_pos(double *p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Stage zeta/path tables into LDS (or return the global pointers).
template <bool LDS_TABLES>
__device__ __forceinline__ void stage_tables(const KArgs &a, unsigned char *smem,
                                             const uint4 *&path_tab, const double *&zeta_tab) {
    if (LDS_TABLES) {
        uint4 *lp = reinterpret_cast<uint4 *>(smem);
        double *lz = reinterpret_cast<double *>(smem + (size_t)a.n_paths * sizeof(uint4));
        for (uint32_t k = threadIdx.x; k < a.n_paths; k += blockDim.x) lp[k] = a.path_rec[k];
        for (uint32_t k = threadIdx.x; k < a.zlen_staged; k += blockDim.x) lz[k] = a.zetas[k];
        __syncthreads();
        path_tab = lp; zeta_tab = lz;
    } else {
        path_tab = a.path_rec; zeta_tab = a.zetas;
    }
}

__device__ __forceinline__ void flush_counters(const KArgs &a, uint32_t done, uint32_t att) {
    // wave64 butterfly, one atomic per wave
    unsigned long long d = done, t = att;
    for (int off = 32; off > 0; off >>= 1) {
        d += __shfl_xor(d, off, 64);
        t += __shfl_xor(t, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&a.counters[0], d);
        atomicAdd(&a.counters[1], t);
    }
}

// ------------------------------------------------------------------------------------------
// K1: 1D
// ------------------------------------------------------------------------------------------
template <bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__global__ void sgd1d_kernel(const KArgs a) {
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
        double *x = a.x;
        while (done < quota && att < max_att) {
            ++att;
            uint4 ra, rb; uint32_t sa, sb, cnt, path;
            if (!sample_pair<LDS_TABLES>(a, path_tab, zeta_tab, rng, ra, rb, sa, sb, cnt, path)) continue;
            double term_dist = fabs(rec_pos(ra) - rec_pos(rb));                        // sgd.rs:513
            if (term_dist == 0.0) continue;                                            // :514
            double mu = fmin(a.it.eta * (1.0 / term_dist), 1.0);                       // :518-520
            const uint32_t i = ra.x, j = rb.x;
            if (i == 0xFFFFFFFFu || j == 0xFFFFFFFFu) continue;                        // :525-538
            double dx;
            if (a.dbg & 2u) dx = (double)i - (double)j;                                // ablation: no position loads
            else dx = load_pos<ATOMIC_LOADS>(x + i) - load_pos<ATOMIC_LOADS>(x + j);   // :541-543
            if (dx == 0.0) dx = 1e-9;                                                  // :546-548
            double mag = fabs(dx);                                                     // :551
            double delta = mu * (mag - term_dist) / 2.0;                               // :552
            double r = delta / mag;                                                    // :570
            double r_x = r * dx;                                                       // :571
            if (a.dbg & 1u) { asm volatile("" :: "v"(r_x)); }                          // ablation: no atomics
            else {
                add_pos(x + i, -r_x);                                                  // :575
                add_pos(x + j, r_x);                                                   // :576
            }
            ++done;                                                                    // :579
            if (TRACE) {
                if (ntr < a.trace_per_stream) {
                    TraceTerm *t = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ntr;
                    t->i = i; t->j = j; t->d = term_dist;
                    ++ntr;
                }
            }
        }
        a.rng[tid] = rng.s0; a.rng[T + tid] = rng.s1; a.rng[2 * T + tid] = rng.s2; a.rng[3 * T + tid] = rng.s3;
        if (TRACE) a.trace_cnt[tid] = ntr;
    }
    flush_counters(a, done, att);
}

// ------------------------------------------------------------------------------------------
// K1b: 1D team kernel — bundled ("run") sampling (sgd_device.h).  A wave is a team:
//   pass   : all 64 lanes sample one leader term each from their own reference streams — the
//            Zipf/f64 arithmetic runs at full SIMD width instead of on one lane per bundle;
//   trips  : B trips execute the 64 leaders as 64/B runs of B lanes.  The records of trip t+1
//            are requested before trip t is consumed, and (DEFER) the atomics of trip t are
//            issued behind the position loads of trip t+1, so a trip exposes one memory round
//            trip (its position loads) instead of three and never waits for its own atomics
//            (vmcnt retires in order: an atomic issued before a load would be waited for).
//            Deferring doubles the window in which a wave reads positions it is about to
//            change; the host enables it only when in-flight terms are few relative to the
//            number of nodes (4*n_streams <= n_nodes) — on small graphs with many streams the
//            extra staleness pushed the concurrent corrections per node past stability.
// The quota is per WAVE with a rank cut-off in the last trip: an iteration performs exactly its
// number of updates; leaders left over when the quota fills are discarded.
// ------------------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ uint32_t bcast(uint32_t v, int leader_lane) {
    if (B == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, leader_lane);     // wave-uniform leader
    return (uint32_t)__shfl((int)v, leader_lane, 64);
}

template <int B, bool DEFER, bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__global__ void sgd1d_team_kernel(const KArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint4 *path_tab; const double *zeta_tab;
    stage_tables<LDS_TABLES>(a, smem, path_tab, zeta_tab);

    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;       // n_streams % 64 == 0 (host-checked)
    if (tid >= a.n_streams) return;                                   // whole waves only
    const int lane = threadIdx.x & 63;
    const int sub = lane & (B - 1);
    const int q = lane / B;
    constexpr int RUNS = 64 / B;                                      // runs per trip
    const uint64_t T = a.n_streams;
    Rng rng;
    rng.s0 = a.rng[tid]; rng.s1 = a.rng[T + tid]; rng.s2 = a.rng[2 * T + tid]; rng.s3 = a.rng[3 * T + tid];
    // wave quota = sum of its 64 lanes' per-stream quotas
    const uint32_t wave_first = tid & ~63u;
    uint64_t wave_quota = (uint64_t)a.quota_base * 64u;
    if (wave_first < a.quota_rem) wave_quota += (a.quota_rem - wave_first) < 64u ? (a.quota_rem - wave_first) : 64u;
    const uint64_t max_passes = (uint64_t)a.attempt_factor * (wave_quota / (64u * B) + 1u) + 16u;
    uint64_t wave_done = 0, passes = 0;
    uint32_t done = 0, att = 0;
    uint32_t ntr = TRACE ? a.trace_cnt[tid] : 0;
    double *x = a.x;
    // deferred atomics of the previous trip
    bool pend = false; uint32_t pend_i = 0, pend_j = 0; double pend_r = 0.0;

    while (wave_done < wave_quota && passes < max_passes) {
        ++passes;
        const Leader L = sample_leader<LDS_TABLES>(a, path_tab, zeta_tab, rng);
        // trip 0: expand and request records
        uint32_t sa = 0, sb = 0;
        bool valid = expand_run<B>(bcast<B>(L.ok, q), bcast<B>(L.first, q), bcast<B>(L.cnt, q),
                                   bcast<B>(L.ra0, q), bcast<B>(L.rb0, q), sub, sa, sb);
        uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
        if (valid) { ra = a.step_rec[sa]; rb = a.step_rec[sb]; }
#pragma unroll 2
        for (int t = 0; t < B; ++t) {
            // request the records of trip t+1
            uint32_t sa_n = 0, sb_n = 0; bool valid_n = false;
            uint4 ra_n = make_uint4(0, 0, 0, 0), rb_n = make_uint4(0, 0, 0, 0);
            if (t + 1 < B) {
                const int ll = (t + 1) * RUNS + q;
                valid_n = expand_run<B>(bcast<B>(L.ok, ll), bcast<B>(L.first, ll), bcast<B>(L.cnt, ll),
                                        bcast<B>(L.ra0, ll), bcast<B>(L.rb0, ll), sub, sa_n, sb_n);
                if (valid_n) { ra_n = a.step_rec[sa_n]; rb_n = a.step_rec[sb_n]; }
            }
            // consume trip t
            ++att;
            double term_dist = 0.0;
            uint32_t i = 0, j = 0;
            if (valid) {
                term_dist = fabs(rec_pos(ra) - rec_pos(rb));                           // sgd.rs:513
                i = ra.x; j = rb.x;
                valid = term_dist != 0.0 && i != 0xFFFFFFFFu && j != 0xFFFFFFFFu;      // :514, :525-538
            }
            const unsigned long long vmask = __ballot(valid);
            const uint64_t remaining = wave_quota - wave_done;
            const uint32_t nvalid = (uint32_t)__popcll(vmask);
            if (valid && nvalid > remaining) {
                const uint32_t rank = (uint32_t)__popcll(vmask & ((1ull << lane) - 1ull));
                valid = rank < remaining;
            }
            wave_done += nvalid < remaining ? nvalid : remaining;
            double xi = 0.0, xj = 0.0;
            if (valid) { xi = load_pos<ATOMIC_LOADS>(x + i); xj = load_pos<ATOMIC_LOADS>(x + j); }   // :541-542
            if (DEFER) {
                if (pend) { add_pos(x + pend_i, -pend_r); add_pos(x + pend_j, pend_r); }   // trip t-1's :575-576
                pend = valid;
            }
            if (valid) {
                double mu = fmin(a.it.eta * (1.0 / term_dist), 1.0);                   // :518-520
                double dx = xi - xj;                                                   // :543
                if (dx == 0.0) dx = 1e-9;                                              // :546-548
                double mag = fabs(dx);                                                 // :551
                double delta = mu * (mag - term_dist) / 2.0;                           // :552
                double r = delta / mag;                                                // :570
                pend_r = r * dx; pend_i = i; pend_j = j;                               // :571
                if (!DEFER) { add_pos(x + i, -pend_r); add_pos(x + j, pend_r); }       // :575-576
                ++done;                                                                // :579
                if (TRACE) {
                    if (ntr < a.trace_per_stream) {
                        TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ntr;
                        tt->i = i; tt->j = j; tt->d = term_dist;
                        ++ntr;
                    }
                }
            }
            if (wave_done >= wave_quota) break;                                        // leaders left over are discarded
            ra = ra_n; rb = rb_n; sa = sa_n; sb = sb_n; valid = valid_n;
        }
    }
    if (pend) { add_pos(x + pend_i, -pend_r); add_pos(x + pend_j, pend_r); }
    a.rng[tid] = rng.s0; a.rng[T + tid] = rng.s1; a.rng[2 * T + tid] = rng.s2; a.rng[3 * T + tid] = rng.s3;
    if (TRACE) a.trace_cnt[tid] = ntr;
    flush_counters(a, done, att);
}

// ------------------------------------------------------------------------------------------
// K2: nD, D compile-time.  coords in Layout order: [node][end][dim] (src/layout.rs:14).
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
            uint4 ra, rb; uint32_t sa, sb, cnt, path;
            if (!sample_pair<LDS_TABLES>(a, path_tab, zeta_tab, rng, ra, rb, sa, sb, cnt, path)) continue;
            const uint32_t first = path_tab[path].x;
            const uint32_t last_step = first + cnt - 1u;
            const uint64_t plen = a.path_len[path];
            uint64_t pa = ((uint64_t)ra.w << 32) | ra.z, pb = ((uint64_t)rb.w << 32) | rb.z;
            uint64_t na, nb;                       // position of the following step / path end
            if (sa == last_step) na = plen; else { uint4 n = a.step_rec[sa + 1u]; na = ((uint64_t)n.w << 32) | n.z; }
            if (sb == last_step) nb = plen; else { uint4 n = a.step_rec[sb + 1u]; nb = ((uint64_t)n.w << 32) | n.z; }
            double pos_a = (double)pa, pos_b = (double)pb;                             // sgd.rs:1047-1048
            const double len_i = (double)(na - pa), len_j = (double)(nb - pb);        // :1051-1058
            const bool rev_i = (ra.y >> 31) != 0, rev_j = (rb.y >> 31) != 0;           // :1061,1070
            bool oa = rng.flip() == 1u;                                                // :1062
            if (oa) { pos_a += len_i; oa = !rev_i; } else { oa = rev_i; }              // :1063-1068
            bool ob = rng.flip() == 1u;                                                // :1071
            if (ob) { pos_b += len_j; ob = !rev_j; } else { ob = rev_j; }              // :1072-1077
            double term_dist = fabs(pos_a - pos_b);                                    // :1080
            if (term_dist == 0.0) continue;                                            // :1081
            double mu = fmin(a.it.eta * (1.0 / term_dist), 1.0);                       // :1085-1086
            if (ra.x == 0xFFFFFFFFu || rb.x == 0xFFFFFFFFu) continue;                  // :1089-1096
            const uint64_t idx_i = (uint64_t)ra.x * 2u + (oa ? 1u : 0u);               // :1099-1103
            const uint64_t idx_j = (uint64_t)rb.x * 2u + (ob ? 1u : 0u);
            double *ci = a.x + idx_i * D, *cj = a.x + idx_j * D;
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

// ------------------------------------------------------------------------------------------
// K2b: nD team kernel — the pass/trip structure of K1b for path_linear_sgd_layout.  Each acting
// lane draws its two end flips (sgd.rs:1062,1071) from its OWN stream; node lengths come from the
// following step record as in K2.  A run of B consecutive nodes is 32*B contiguous coordinate
// bytes for D = 2 (Layout order), so coordinate loads and atomics coalesce like the 1D positions.
// Atomics are issued in the trip that computes them (no deferral).
// ------------------------------------------------------------------------------------------
template <int D, int B, bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__global__ void sgdnd_team_kernel(const KArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint4 *path_tab; const double *zeta_tab;
    stage_tables<LDS_TABLES>(a, smem, path_tab, zeta_tab);

    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= a.n_streams) return;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (B - 1);
    const int q = lane / B;
    constexpr int RUNS = 64 / B;
    const uint64_t T = a.n_streams;
    Rng rng;
    rng.s0 = a.rng[tid]; rng.s1 = a.rng[T + tid]; rng.s2 = a.rng[2 * T + tid]; rng.s3 = a.rng[3 * T + tid];
    const uint32_t wave_first = tid & ~63u;
    uint64_t wave_quota = (uint64_t)a.quota_base * 64u;
    if (wave_first < a.quota_rem) wave_quota += (a.quota_rem - wave_first) < 64u ? (a.quota_rem - wave_first) : 64u;
    const uint64_t max_passes = (uint64_t)a.attempt_factor * (wave_quota / (64u * B) + 1u) + 16u;
    uint64_t wave_done = 0, passes = 0;
    uint32_t done = 0, att = 0;
    uint32_t ntr = TRACE ? a.trace_cnt[tid] : 0;

    while (wave_done < wave_quota && passes < max_passes) {
        ++passes;
        const Leader L = sample_leader<LDS_TABLES>(a, path_tab, zeta_tab, rng);
        uint32_t sa = 0, sb = 0;
        uint32_t first = bcast<B>(L.first, q), cnt = bcast<B>(L.cnt, q);
        bool valid = expand_run<B>(bcast<B>(L.ok, q), first, cnt, bcast<B>(L.ra0, q), bcast<B>(L.rb0, q), sub, sa, sb);
        uint4 ra = make_uint4(0, 0, 0, 0), rb = ra, na = ra, nb = ra;
        if (valid) {
            ra = a.step_rec[sa]; rb = a.step_rec[sb];
            na = a.step_rec[sa + 1u < a.n_steps ? sa + 1u : sa]; nb = a.step_rec[sb + 1u < a.n_steps ? sb + 1u : sb];
        }
#pragma unroll 2
        for (int t = 0; t < B; ++t) {
            uint32_t sa_n = 0, sb_n = 0, first_n = 0, cnt_n = 0; bool valid_n = false;
            uint4 ra_n = make_uint4(0, 0, 0, 0), rb_n = ra_n, na_n = ra_n, nb_n = ra_n;
            if (t + 1 < B) {
                const int ll = (t + 1) * RUNS + q;
                first_n = bcast<B>(L.first, ll); cnt_n = bcast<B>(L.cnt, ll);
                valid_n = expand_run<B>(bcast<B>(L.ok, ll), first_n, cnt_n, bcast<B>(L.ra0, ll), bcast<B>(L.rb0, ll), sub, sa_n, sb_n);
                if (valid_n) {
                    ra_n = a.step_rec[sa_n]; rb_n = a.step_rec[sb_n];
                    na_n = a.step_rec[sa_n + 1u < a.n_steps ? sa_n + 1u : sa_n];
                    nb_n = a.step_rec[sb_n + 1u < a.n_steps ? sb_n + 1u : sb_n];
                }
            }
            ++att;
            double term_dist = 0.0;
            uint64_t idx_i = 0, idx_j = 0;
            if (valid) {
                const uint32_t last_step = first + cnt - 1u;
                const uint64_t plen = a.path_len[ra.y & 0x7FFFFFFFu];
                const uint64_t pa = ((uint64_t)ra.w << 32) | ra.z, pb = ((uint64_t)rb.w << 32) | rb.z;
                const uint64_t ea = sa == last_step ? plen : (((uint64_t)na.w << 32) | na.z);
                const uint64_t eb = sb == last_step ? plen : (((uint64_t)nb.w << 32) | nb.z);
                double pos_a = (double)pa, pos_b = (double)pb;                         // sgd.rs:1047-1048
                const double len_i = (double)(ea - pa), len_j = (double)(eb - pb);     // :1051-1058
                const bool rev_i = (ra.y >> 31) != 0, rev_j = (rb.y >> 31) != 0;
                bool oa = rng.flip() == 1u;                                            // :1062
                if (oa) { pos_a += len_i; oa = !rev_i; } else { oa = rev_i; }
                bool ob = rng.flip() == 1u;                                            // :1071
                if (ob) { pos_b += len_j; ob = !rev_j; } else { ob = rev_j; }
                term_dist = fabs(pos_a - pos_b);                                       // :1080
                valid = term_dist != 0.0 && ra.x != 0xFFFFFFFFu && rb.x != 0xFFFFFFFFu;   // :1081, :1089-1096
                idx_i = (uint64_t)ra.x * 2u + (oa ? 1u : 0u);                          // :1099-1103
                idx_j = (uint64_t)rb.x * 2u + (ob ? 1u : 0u);
            }
            const unsigned long long vmask = __ballot(valid);
            const uint64_t remaining = wave_quota - wave_done;
            const uint32_t nvalid = (uint32_t)__popcll(vmask);
            if (valid && nvalid > remaining) {
                const uint32_t rank = (uint32_t)__popcll(vmask & ((1ull << lane) - 1ull));
                valid = rank < remaining;
            }
            wave_done += nvalid < remaining ? nvalid : remaining;
            if (valid) {
                double mu = fmin(a.it.eta * (1.0 / term_dist), 1.0);                   // :1085-1086
                double *ci = a.x + idx_i * D, *cj = a.x + idx_j * D;
                double deltas[D];
                double mag_sq = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) {                                          // :1108-1113
                    deltas[d] = load_pos<ATOMIC_LOADS>(ci + d) - load_pos<ATOMIC_LOADS>(cj + d);
                    mag_sq += deltas[d] * deltas[d];
                }
                if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }               // :1116-1119
                double mag = sqrt(mag_sq);                                             // :1121
                double delta = mu * (mag - term_dist) / 2.0;                           // :1125
                double r = delta / mag;                                                // :1142
                const bool same = idx_i == idx_j;
#pragma unroll
                for (int d = 0; d < D; ++d) {                                          // :1143-1149
                    double r_d = r * deltas[d];
                    if (!same) add_pos(ci + d, -r_d);
                    add_pos(cj + d, r_d);
                }
                ++done;                                                                // :1151
                if (TRACE) {
                    if (ntr < a.trace_per_stream) {
                        TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ntr;
                        tt->i = (uint32_t)idx_i; tt->j = (uint32_t)idx_j; tt->d = term_dist;
                        ++ntr;
                    }
                }
            }
            if (wave_done >= wave_quota) break;
            ra = ra_n; rb = rb_n; na = na_n; nb = nb_n; sa = sa_n; sb = sb_n; valid = valid_n; first = first_n; cnt = cnt_n;
        }
    }
    a.rng[tid] = rng.s0; a.rng[T + tid] = rng.s1; a.rng[2 * T + tid] = rng.s2; a.rng[3 * T + tid] = rng.s3;
    if (TRACE) a.trace_cnt[tid] = ntr;
    flush_counters(a, done, att);
}

// ---- host-side launch dispatch -------------------------------------------------------------
template <bool L, bool A>
static hipError_t launch_1d_t(const KArgs &a, bool trace, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (trace) hipLaunchKernelGGL((sgd1d_kernel<L, A, true>), grid, block, lds, st, a);
    else       hipLaunchKernelGGL((sgd1d_kernel<L, A, false>), grid, block, lds, st, a);
    return hipGetLastError();
}
template <int B, bool L, bool A>
static hipError_t launch_1db_t(const KArgs &a, bool trace, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    const bool defer = (a.dbg & 0x80u) != 0;          // host decision, see capi.hip
    if (defer) {
        if (trace) hipLaunchKernelGGL((sgd1d_team_kernel<B, true, L, A, true>), grid, block, lds, st, a);
        else       hipLaunchKernelGGL((sgd1d_team_kernel<B, true, L, A, false>), grid, block, lds, st, a);
    } else {
        if (trace) hipLaunchKernelGGL((sgd1d_team_kernel<B, false, L, A, true>), grid, block, lds, st, a);
        else       hipLaunchKernelGGL((sgd1d_team_kernel<B, false, L, A, false>), grid, block, lds, st, a);
    }
    return hipGetLastError();
}
template <int B>
static hipError_t launch_1db(const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                             dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (lds_tables) return atomic_loads ? launch_1db_t<B, true, true>(a, trace, grid, block, lds, st)
                                        : launch_1db_t<B, true, false>(a, trace, grid, block, lds, st);
    return atomic_loads ? launch_1db_t<B, false, true>(a, trace, grid, block, 0, st)
                        : launch_1db_t<B, false, false>(a, trace, grid, block, 0, st);
}
hipError_t launch_1d(const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                     dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    switch (a.bundle) {
        case 0: case 1:
            if (lds_tables) return atomic_loads ? launch_1d_t<true, true>(a, trace, grid, block, lds, st)
                                                : launch_1d_t<true, false>(a, trace, grid, block, lds, st);
            return atomic_loads ? launch_1d_t<false, true>(a, trace, grid, block, 0, st)
                                : launch_1d_t<false, false>(a, trace, grid, block, 0, st);
        case 4:  return launch_1db<4>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 8:  return launch_1db<8>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 16: return launch_1db<16>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 32: return launch_1db<32>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 64: return launch_1db<64>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        default: return hipErrorInvalidValue;
    }
}

template <int D, bool L, bool A>
static hipError_t launch_nd_t(const KArgs &a, bool trace, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (trace) hipLaunchKernelGGL((sgdnd_kernel<D, L, A, true>), grid, block, lds, st, a);
    else       hipLaunchKernelGGL((sgdnd_kernel<D, L, A, false>), grid, block, lds, st, a);
    return hipGetLastError();
}
template <int D, int B>
static hipError_t launch_ndb(const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                             dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    // team kernel variants: LDS tables on/off; agent-scope loads; trace on/off
    if (lds_tables) {
        if (trace) hipLaunchKernelGGL((sgdnd_team_kernel<D, B, true, true, true>), grid, block, lds, st, a);
        else if (atomic_loads) hipLaunchKernelGGL((sgdnd_team_kernel<D, B, true, true, false>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((sgdnd_team_kernel<D, B, true, false, false>), grid, block, lds, st, a);
    } else {
        if (trace) hipLaunchKernelGGL((sgdnd_team_kernel<D, B, false, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((sgdnd_team_kernel<D, B, false, true, false>), grid, block, 0, st, a);
    }
    return hipGetLastError();
}
template <int D>
static hipError_t launch_nd_d(const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                              dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (D <= 3) {   // bundled layout kernels are built for the common dimensions only
        switch (a.bundle) {
            case 8:  return launch_ndb<(D <= 3 ? D : 1), 8>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
            case 16: return launch_ndb<(D <= 3 ? D : 1), 16>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
            case 32: return launch_ndb<(D <= 3 ? D : 1), 32>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
            case 64: return launch_ndb<(D <= 3 ? D : 1), 64>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
            default: break;
        }
    }
    if (lds_tables) return atomic_loads ? launch_nd_t<D, true, true>(a, trace, grid, block, lds, st)
                                        : launch_nd_t<D, true, false>(a, trace, grid, block, lds, st);
    return atomic_loads ? launch_nd_t<D, false, true>(a, trace, grid, block, 0, st)
                        : launch_nd_t<D, false, false>(a, trace, grid, block, 0, st);
}
hipError_t launch_nd(int dims, const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                     dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    switch (dims) {
        case 1: return launch_nd_d<1>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 2: return launch_nd_d<2>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 3: return launch_nd_d<3>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 4: return launch_nd_d<4>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 5: return launch_nd_d<5>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 6: return launch_nd_d<6>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 7: return launch_nd_d<7>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        case 8: return launch_nd_d<8>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace gfs
