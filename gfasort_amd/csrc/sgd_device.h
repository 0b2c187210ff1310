// sgd_device.h — device-side building blocks of the path-guided SGD kernels (gfx950).
//
// Everything here must be BIT-EXACT against the reference arithmetic (src/sgd.rs), so:
//   * compiled with -ffp-contract=off (Rust never fuses a*b+c), no fast-math;
//   * f64 -> integer casts go through explicit saturating helpers (Rust `as` semantics);
//   * f64 division / sqrt are the IEEE-correct device forms (hipcc default without fast-math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gfs {

// ---- Rust `as` casts (sgd.rs:149,157,164) ------------------------------------------------
__device__ __forceinline__ int32_t sat_i32(double v) {
    if (v != v) return 0;
    if (v <= -2147483648.0) return INT32_MIN;
    if (v >= 2147483647.0) return INT32_MAX;
    return (int32_t)v;
}

// ---- fast_precise_pow (sgd.rs:155-182) with the exponent b pre-split on the host into its
// saturated integer part e and the remainder fb = b - (double)e; b is launch-uniform for every
// call site, so the square-and-multiply loop is wave-uniform.
__device__ __forceinline__ double fpp_split(double a, int32_t e, double fb) {
    int32_t high = (int32_t)((uint64_t)__double_as_longlong(a) >> 32);                 // :162-163
    int32_t diff = (int32_t)((uint32_t)high - 1072632447u);                            // wrapping i32
    int32_t new_high = sat_i32(fb * (double)diff + 1072632447.0);                      // :164
    double frac = __longlong_as_double((long long)(((uint64_t)(uint32_t)new_high) << 32));  // :166-167
    double base = a, r = 1.0;
    int32_t ex = e;
    while (ex != 0) {                                                                  // :173-179
        if (ex & 1) r *= base;
        base *= base;
        ex >>= 1;
    }
    return r * frac;                                                                   // :181
}

// ---- Xoshiro256+ (rand_xoshiro 0.7), one generator per lane, state in registers ---------
struct Rng {
    uint64_t s0, s1, s2, s3;
    __device__ __forceinline__ uint64_t next() {
        uint64_t result = s0 + s3;
        uint64_t t = s1 << 17;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 45) | (s3 >> 19);
        return result;
    }
    // rand 0.9 Uniform<i32>(0,2): top bit of one u32 draw (sgd.rs:436,456,460)
    __device__ __forceinline__ uint32_t flip() { return (uint32_t)(next() >> 63); }
    // rng.random::<f64>() (sgd.rs:136)
    __device__ __forceinline__ double f64() {
        return (double)(next() >> 11) * (1.0 / 9007199254740992.0);
    }
    // rand 0.9 Uniform<usize>(0,n) for n <= u32::MAX: u32 draws, widening multiply, accept iff
    // lo >= thresh with thresh = (2^32 - n) mod n precomputed (sgd.rs:435,444,493-494)
    __device__ __forceinline__ uint32_t uniform32(uint32_t n, uint32_t thresh) {
        for (;;) {
            uint64_t m = (uint64_t)(uint32_t)(next() >> 32) * (uint64_t)n;
            if ((uint32_t)m >= thresh) return (uint32_t)(m >> 32);
        }
    }
};

// ---- launch-uniform constants of one SGD batch (host-computed, bit-exact) ----------------
struct IterConsts {
    double   eta;          // etas[k]                                        sgd.rs:389,519
    double   zeta2theta;   // 1.0 + fpp(0.5, theta_cur)  (also the 2nd fast-path bound) :471,143
    double   omt_fb;       // (1 - theta_cur) split for fpp(2/n, 1-theta)               :133
    double   alpha_fb;     // alpha = 1/(1-theta_cur) split for fpp(.., alpha)          :132,148
    int32_t  omt_e;
    int32_t  alpha_e;
    int32_t  cooling;      // k > first_cooling_iteration                               :393-396
    int32_t  _pad;
};

// device mirror of PathIndex (sgd.rs:14-31), flattened:
//   step_rec[s]  = { node dense idx | NO_NODE, path id | rev<<31, pos lo, pos hi }  (16 B)
//   path_rec[p]  = { first_step, step_count, (2^32-count) mod count, 0 }            (16 B)
struct KArgs {
    const uint4    *step_rec;
    const uint4    *path_rec;
    const uint64_t *path_len;      // bp length per path (nD: length of a path's last node)
    const double   *zetas;
    double         *x;             // 1D: x[n_nodes]; nD: coords[n_nodes*2*D] Layout order
    uint64_t       *rng;           // [4][n_streams] SoA
    unsigned long long *counters;  // [0] successful updates, [1] attempts
    void           *trace;         // gfs_term[n_streams*trace_per_stream] or null
    uint32_t       *trace_cnt;     // [n_streams]
    uint32_t n_steps, steps_thresh;
    uint32_t n_paths, zlen_full;   // zlen_full: true table length (index clamp, sgd.rs:469)
    uint32_t zlen_staged;          // entries copied to LDS (>= every reachable index)
    uint32_t n_streams;
    uint32_t quota_base, quota_rem;
    uint32_t attempt_factor, trace_per_stream;
    uint32_t space, space_max, space_q;
    uint32_t dbg;                  // diagnostic ablation bits (GFS_F_DBG_* >> 8), 0 in production
    uint32_t bundle, _pad2;        // lanes per sampling bundle (1 = reference streams)
    IterConsts it;
};

// zeta index rule (sgd.rs:463-469)
__device__ __forceinline__ uint32_t space_index(const KArgs &a, uint32_t jump) {
    uint32_t idx = jump > a.space_max ? a.space_max + (jump - a.space_max) / a.space_q + 1 : jump;
    uint32_t last = a.zlen_full - 1;
    return idx < last ? idx : last;
}

// DirtyZipfian::sample with min = 1, max = n = jump (sgd.rs:128-150); returns z_i.
__device__ __forceinline__ uint32_t dirty_zipf(const IterConsts &c, uint32_t jump, double zeta, double u) {
    double uz = u * zeta;                                                              // :137
    if (uz < 1.0) return 1u;                                                           // :140
    if (uz < c.zeta2theta) return 2u;                                                  // :143 (not clamped)
    double n = (double)jump;
    double eta = (1.0 - fpp_split(2.0 / n, c.omt_e, c.omt_fb)) / (1.0 - c.zeta2theta / zeta);   // :133-134
    double v = eta * u - eta + 1.0;
    double result = 1.0 + n * fpp_split(v, c.alpha_e, c.alpha_fb);                     // :148
    // (result as u64).min(max): saturating cast; max = jump < 2^32
    if (!(result > 0.0)) return 0u;
    if (result >= 4294967296.0) return jump;
    uint32_t r = (uint32_t)result;
    return r < jump ? r : jump;                                                        // :149
}

// One trip of the pair sampler, sgd.rs:444-499 == :990-1037.  Returns false on `continue`.
template <bool LDS_TABLES>
__device__ __forceinline__ bool sample_pair(const KArgs &a, const uint4 *path_tab, const double *zeta_tab,
                                            Rng &rng, uint4 &ra, uint4 &rb, uint32_t &sa, uint32_t &sb,
                                            uint32_t &cnt_out, uint32_t &path_out) {
    uint32_t step_idx = rng.uniform32(a.n_steps, a.steps_thresh);                      // :444
    ra = a.step_rec[step_idx];
    uint32_t path = ra.y & 0x7FFFFFFFu;                                                // :445
    uint4 pr = path_tab[path];
    uint32_t first = pr.x, cnt = pr.y;                                                 // :446
    if (cnt == 1u) return false;                                                       // :448
    uint32_t rank_a = step_idx - first;                                                // :452
    uint32_t rank_b = rank_a;
    if (a.it.cooling || rng.flip() == 1u) {                                            // :456
        bool back = false, fwd = false;
        if (rank_a > 0u && (rng.flip() == 1u || rank_a == cnt - 1u)) back = true;      // :460
        else if (rank_a < cnt - 1u) fwd = true;                                        // :475
        if (back || fwd) {
            uint32_t room = back ? rank_a : (cnt - rank_a - 1u);
            uint32_t jump = a.space < room ? a.space : room;                           // :462,477
            double zeta = zeta_tab[space_index(a, jump)];
            uint32_t z = dirty_zipf(a.it, jump, zeta, rng.f64());                      // :472-473
            if (back) rank_b = rank_a >= z ? rank_a - z : 0u;                          // :474
            else { uint64_t t = (uint64_t)rank_a + z; rank_b = t < cnt - 1u ? (uint32_t)t : cnt - 1u; }  // :489
        }
    } else {
        rank_b = rng.uniform32(cnt, pr.z);                                             // :493-494
    }
    if (rank_a == rank_b) return false;                                                // :497
    sa = step_idx;                                                                     // :502
    sb = first + rank_b;                                                               // :503
    rb = a.step_rec[sb];
    cnt_out = cnt; path_out = path;
    return true;
}

// ------------------------------------------------------------------------------------------
// Bundled sampler ("run sampling", GFS_F_BUNDLE(B)): B adjacent lanes form a bundle.  The
// bundle's first lane (the leader) is an ordinary reference stream: it draws step a0, the
// Zipf/uniform decision, the direction and the jump exactly as sgd.rs:444-495 does.  Lane l of
// the bundle (a satellite) takes the step l places further along the path (wrapping to the path
// start) and the SAME signed jump:  rank_a = (rank_a0 + l) mod cnt,  rank_b = rank_a + shift.
//   * a satellite whose rank_b falls outside the path is rejected (never clamped: clamping
//     would pile several lanes onto the path's last step);
//   * when |shift| < B the bundle's terms would chain through shared nodes (a_{l+z} = b_l);
//     only lanes with floor(l/z) even act, which makes the acting terms node-disjoint;
//   * paths shorter than 2B steps are handled by the leader alone.
// Every acting lane's term has the reference's marginal distribution up to path-end effects of
// O(B/cnt); what changes is the correlation BETWEEN concurrent terms.  The pay-off is in the
// memory system: B consecutive steps are 16*B contiguous record bytes and (for a locally sorted
// graph) B neighbouring position words, so record loads, position loads and the f64 atomics of
// a bundle coalesce into a few 64-B requests instead of B scattered ones.
// ------------------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ bool sample_pair_bundled(const KArgs &a, const uint4 *path_tab, const double *zeta_tab,
                                                    Rng &rng, uint4 &ra, uint4 &rb, uint32_t &sa, uint32_t &sb,
                                                    uint32_t &first_out, uint32_t &cnt_out, uint32_t &path_out) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & (B - 1);
    const int lead = lane & ~(B - 1);
    uint32_t s0 = 0;
    if (sub == 0) s0 = rng.uniform32(a.n_steps, a.steps_thresh);                       // sgd.rs:444 (leader)
    s0 = __shfl(s0, lead, 64);
    // speculative, coalesced load of the bundle's a-records (valid unless the run leaves the path)
    uint32_t s_spec = s0 + (uint32_t)sub;
    if (s_spec >= a.n_steps) s_spec = a.n_steps - 1u;
    ra = a.step_rec[s_spec];
    const uint32_t path = __shfl(ra.y, lead, 64) & 0x7FFFFFFFu;                        // :445
    const uint4 pr = path_tab[path];
    const uint32_t first = pr.x, cnt = pr.y;                                           // :446
    if (cnt == 1u) return false;                                                       // :448 (whole bundle)
    const uint32_t rank_a0 = s0 - first;                                               // :452
    uint32_t rank_b0 = rank_a0;
    if (sub == 0) {
        if (a.it.cooling || rng.flip() == 1u) {                                        // :456
            bool back = false, fwd = false;
            if (rank_a0 > 0u && (rng.flip() == 1u || rank_a0 == cnt - 1u)) back = true;  // :460
            else if (rank_a0 < cnt - 1u) fwd = true;                                   // :475
            if (back || fwd) {
                uint32_t room = back ? rank_a0 : (cnt - rank_a0 - 1u);
                uint32_t jump = a.space < room ? a.space : room;                       // :462,477
                double zeta = zeta_tab[space_index(a, jump)];
                uint32_t z = dirty_zipf(a.it, jump, zeta, rng.f64());                  // :472-473
                if (back) rank_b0 = rank_a0 >= z ? rank_a0 - z : 0u;                   // :474
                else { uint64_t t = (uint64_t)rank_a0 + z; rank_b0 = t < cnt - 1u ? (uint32_t)t : cnt - 1u; }  // :489
            }
        } else {
            rank_b0 = rng.uniform32(cnt, pr.z);                                        // :493-494
        }
    }
    rank_b0 = __shfl(rank_b0, lead, 64);
    if (rank_b0 == rank_a0) return false;                                              // :497 (leader's term; bundle follows)
    uint32_t ra_l = rank_a0, rb_l = rank_b0;
    if (sub != 0) {
        if (cnt < 2u * B) return false;                                                // short path: leader only
        const int64_t shift = (int64_t)rank_b0 - (int64_t)rank_a0;
        const uint32_t z = (uint32_t)(shift < 0 ? -shift : shift);
        if (z < (uint32_t)B && ((((uint32_t)sub) / z) & 1u)) return false;             // node-disjoint lanes only
        ra_l = rank_a0 + (uint32_t)sub;
        const bool wrapped = ra_l >= cnt;
        if (wrapped) ra_l -= cnt;
        const int64_t t = (int64_t)ra_l + shift;
        if (t < 0 || t > (int64_t)cnt - 1) return false;                               // outside the path: reject
        rb_l = (uint32_t)t;
        if (wrapped) ra = a.step_rec[first + ra_l];                                    // rare: reload
    }
    sa = first + ra_l;                                                                 // :502
    sb = first + rb_l;                                                                 // :503
    rb = a.step_rec[sb];
    first_out = first; cnt_out = cnt; path_out = path;
    return true;
}

__device__ __forceinline__ double rec_pos(const uint4 &r) {
    return (double)(((uint64_t)r.w << 32) | (uint64_t)r.z);
}

}  // namespace gfs
