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
    // ... and for n > u32::MAX: u64 draws, 128-bit widening multiply, thresh = (2^64 - n) mod n
    __device__ __forceinline__ uint64_t uniform64(uint64_t n, uint64_t thresh) {
        for (;;) {
            const uint64_t r = next();
            if (r * n >= thresh) return __umul64hi(r, n);
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
//   step_rec[s]  = { node slot | NO_NODE, path id (31 bits) | rev<<31, pos lo, pos hi (23 bits) | crowding a<<23 b<<29 }  (16 B)
//   path_rec[p]  = { first_step lo, step_count, (2^32-count) mod count, first_step hi } (16 B)
struct KArgs {
    const uint4    *step_rec;
    const uint4    *path_rec;
    const uint64_t *path_len;      // bp length per path (nD: length of a path's last node)
    const double   *zetas;
    double         *x;             // 1D: x[slot]; nD: planes coords[end][dim][slot] (see coord_ptr)
    uint64_t       *rng;           // [4][n_streams] SoA
    unsigned long long *counters;  // [slots][8]: [s][0] successful updates, [s][1] attempts (sgd_kernel_common.h)
    void           *trace;         // gfs_term[n_streams*trace_per_stream] or null
    uint32_t       *trace_cnt;     // [n_streams]
    uint32_t       *lead;          // [8][n_streams] SoA: leaders a 1D team wave has sampled but not yet expanded
                                   // (first lo, first hi, cnt, ra0, rb0, ok | trips left << 8 | ..., ra1, rb1: load_pass); or null
    uint64_t n_steps, steps_thresh;   // thresh = (2^w - n) mod n, w = 32 if n_steps <= u32::MAX else 64
    uint32_t n_paths, zlen_full;   // zlen_full: true table length (index clamp, sgd.rs:469)
    uint32_t zlen_staged;          // entries copied to LDS (>= every reachable index)
    uint32_t n_streams;
    uint32_t quota_base, quota_rem;
    uint32_t attempt_factor, trace_per_stream;
    uint32_t space, space_max, space_q;
    uint32_t dbg;                  // diagnostic ablation bits (GFS_F_DBG_* >> 8), 0 in production
    uint32_t bundle, n_nodes;      // lanes per sampling bundle (1 = reference streams); node count (nD planes)
    int32_t  kshift;               // crowding: floor(log2(n_steps / (2 * n_streams))) + 2, see crowd_shift()
    uint32_t chain;                // longest run in trips (power of two; 1 = a run is one trip), see run_trips()
    uint32_t partners;             // partner draws per leader (1 or 2; 2 only for the 1D team kernel at B = 64), see Leader
    uint32_t ref_chunk;            // K1d / K2d: updates per live lane and pool claim (sgd_kernel_common.h ref_pooled_walk)
    uint32_t dbg2;                 // diagnostic bits for experiment builds (GFS_DBG2 environment variable); no kernel reads them at present
    uint32_t chunk;                // updates per chunk of a team wave's work (TEAM_CHUNK; the probe knob GFS_DBG_ONE_CHUNK changes it)
    IterConsts it;
};

// step_idx ~ U[0, total_steps) (sgd.rs:444): rand's usize sampler draws u32 while the range fits
// u32 and u64 beyond — graphs past 2^32 steps fit a 288 GB MI355X (16 B per step record).
__device__ __forceinline__ uint64_t sample_step(const KArgs &a, Rng &rng) {
    if (a.n_steps <= 0xFFFFFFFFull && !(a.dbg & 0x40u)) return rng.uniform32((uint32_t)a.n_steps, (uint32_t)a.steps_thresh);
    return rng.uniform64(a.n_steps, a.steps_thresh);
}
__device__ __forceinline__ uint64_t path_first(const uint4 &pr) { return ((uint64_t)pr.w << 32) | pr.x; }
constexpr uint32_t PATH_MASK = 0x7FFFFFFFu;              // step_rec.y bits 0..30: path id (bit 31: reverse step)
constexpr uint32_t POS_HI_MASK = 0x7FFFFFu;               // step_rec.w bits 0..22: bits 32..54 of the bp position
__device__ __forceinline__ uint64_t rec_pos_u64(const uint4 &r) { return ((uint64_t)(r.w & POS_HI_MASK) << 32) | (uint64_t)r.z; }
__device__ __forceinline__ uint32_t rec_path(const uint4 &r) { return r.y & PATH_MASK; }

// Crowded nodes.  The kernels run ~2.5e5 terms at once where the reference runs <= 64, and a term corrects its two
// nodes from positions read before the other in-flight terms landed.  For an ordinary node that is at most one or two
// concurrent corrections (the streams-per-node bound, capi.hip); but a node that carries a large share of all steps
// (a hub), or that a path steps on many times in a row (a tandem repeat: a run of 64 consecutive steps then hits it
// with many lanes of the SAME trip), receives dozens of full corrections of the same error at once and the positions
// blow up (measured: NaN on graphs with 40-fold self-loops, profiles/r01/repeat_probe.log).  Every step record
// therefore carries two small exponents of its node (in the spare top bits of its position's high word: bp positions
// stay below 2^55), computed when the index is built (index_kernels.hip):
//   a = ceil(log2(steps on the node)),  b = ceil(log2(most visits within any 64 consecutive steps of a path)),
// and a term's mu is scaled by 2^-k, k = max over its two nodes of max(b, a - kshift): c concurrent corrections of
// 1/c-th size add up to about one.  kshift = floor(log2(n_steps / (2 n_streams))) + 2 puts the onset at four times
// the concurrency an average node sees, so ordinary graphs (every test graph of the parity ladder) have k = 0
// everywhere and are bit-for-bit unaffected.  The reference has no such rule; it has no such concurrency either.
template <bool RUNS>   // RUNS: the lanes of a wave take consecutive steps (team kernels), so tandem repeats matter
__device__ __forceinline__ int crowd_shift(const KArgs &a, const uint4 &ra, const uint4 &rb) {
    const int aa = (int)((ra.w >> 23) & 63u) - a.kshift, ab = (int)((rb.w >> 23) & 63u) - a.kshift;
    const int ba = RUNS ? (int)((ra.w >> 29) & 7u) : 0, bb = RUNS ? (int)((rb.w >> 29) & 7u) : 0;
    int k = aa > ab ? aa : ab;
    const int b = ba > bb ? ba : bb;
    k = k > b ? k : b;
    return k > 0 ? k : 0;
}
__device__ __forceinline__ double crowd_scale(double mu, int k) { return k ? ldexp(mu, -k) : mu; }

// nD coordinates on the device: END x DIMENSION PLANES, coords[end][dim][slot].  (The ABI and Layout.coords, layout.rs:14,
// are [node][end][dim]; upload/download translate.)  A run of consecutive nodes taking the same end then reads and adds
// 8*B CONTIGUOUS bytes per dimension, lane l the l-th of them: every instruction of a team wave is 8 fully used lines, and
// nothing has to be re-dealt between lanes first.  (Round 2 kept [end][slot][dim]: the same request count, but only after
// 12-24 lane permutes per trip, ~12 % of the layout kernel's time — profiles/r03/nd_ablate.log.)  A lone term of a reference
// stream issues one request per end and dimension in either order.
// coord_ptr: dimension 0 of an end; coord_step: elements between its dimensions.
template <int D>
__device__ __forceinline__ double *coord_ptr(const KArgs &a, uint32_t slot, bool end) {
    return a.x + (end ? (uint64_t)a.n_nodes * D : 0ull) + slot;
}
__device__ __forceinline__ uint64_t coord_step(const KArgs &a) { return (uint64_t)a.n_nodes; }

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

// One trip of the pair sampler, sgd.rs:444-499 == :990-1037, from a step a that has been drawn (sgd.rs:444) and whose record
// has been requested already.  Returns false on `continue`.
template <bool LDS_TABLES>
__device__ __forceinline__ bool sample_pair_from(const KArgs &a, const uint4 *path_tab, const double *zeta_tab,
                                                 Rng &rng, const uint64_t step_idx, const uint4 &ra, uint4 &rb, uint64_t &sa, uint64_t &sb,
                                                 uint32_t &cnt_out, uint32_t &path_out) {
    uint32_t path = rec_path(ra);                                                      // :445
    uint4 pr = path_tab[path];
    const uint64_t first = path_first(pr); const uint32_t cnt = pr.y;                  // :446
    if (cnt == 1u) return false;                                                       // :448
    uint32_t rank_a = (uint32_t)(step_idx - first);                                    // :452
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
template <bool LDS_TABLES>
__device__ __forceinline__ bool sample_pair(const KArgs &a, const uint4 *path_tab, const double *zeta_tab,
                                            Rng &rng, uint4 &ra, uint4 &rb, uint64_t &sa, uint64_t &sb,
                                            uint32_t &cnt_out, uint32_t &path_out) {
    const uint64_t step_idx = sample_step(a, rng);                                     // :444
    ra = a.step_rec[step_idx];
    return sample_pair_from<LDS_TABLES>(a, path_tab, zeta_tab, rng, step_idx, ra, rb, sa, sb, cnt_out, path_out);
}

// ------------------------------------------------------------------------------------------
// Bundled ("run") sampling, GFS_F_BUNDLE(B), B in {4,8,16,32,64} — used by the team kernels.
//
// Every lane is an ordinary reference stream.  In a SAMPLING PASS each lane draws one term
// exactly as sgd.rs:444-497 does — step a0, Zipf/uniform decision, direction, jump — but does
// not apply it: the term becomes the LEADER of a run.  The wave then executes the 64 leaders of
// a pass in B TRIPS; in trip t, bundle q (lanes q*B .. q*B+B-1) expands leader number
// t*(64/B)+q: lane l of the bundle takes the step l places further along the path (wrapping to
// the path start) with the leader's signed jump:
//        rank_a = (rank_a0 + l) mod cnt,      rank_b = rank_a + (rank_b0 - rank_a0).
//   * a lane whose rank_b falls outside the path mirrors the jump to the other side (a step near
//     a path end samples into the path in the reference too) and is rejected if that fails as
//     well — never clamped: clamping would pile several lanes onto the path's last step;
//     lane 0 is the leader's own term, unchanged;
//   * when |jump| < B the run's terms would chain through shared nodes (a_{l+z} = b_l); the run is
//     executed as two node-disjoint trips (colours): lanes with floor(l/z) even, then the odd ones;
//   * paths shorter than 2B steps are handled by lane 0 alone.
// Every acting lane's term has the reference's marginal distribution up to path-end effects of
// O(B/cnt); what changes is the correlation BETWEEN concurrent terms.  The pay-off is in the
// memory system: B consecutive steps are 16*B contiguous record bytes and (in a locally sorted
// graph) B neighbouring position words, so the record loads, position loads and f64 atomics of
// a run coalesce into a few 64-B requests instead of B scattered ones — and since a pass fixes
// the next B trips in advance, their record loads can be issued a trip ahead.
// ------------------------------------------------------------------------------------------
struct Leader {            // one sampled leader (per lane, registers)
    uint32_t first_lo, first_hi, cnt;   // PathInfo of its path (first_step is 64-bit)
    uint32_t ra0, rb0;     // ranks of step a and step b (first partner)
    uint32_t ok;           // bit 0: 0 = the reference `continue`d (cnt == 1 or rank_a == rank_b);
                           // bit 1: both runs line-aligned, bits 2..4: lane rotation r (draw_partner);
                           // bits 8..12: the same for the second partner
    uint32_t ra1, rb1;     // TWO PARTNERS (KArgs.partners = 2): the same step a with a second, independent draw of step b
};
__device__ __forceinline__ uint32_t leader_ok(uint32_t okw, uint32_t p) { return (p ? okw >> 8 : okw) & 0x1Fu; }

// TWO PARTNERS.  With KArgs.partners = 2 a leader is one draw of step a (sgd.rs:444-453) and TWO independent draws of its
// partner b (sgd.rs:456-497), each from the reference's distribution given a: two terms that share their a-side.  Every
// term keeps the reference's marginal distribution; once more only the correlation between terms changes (an a-run now
// takes two corrections in a row, from two unrelated places of its path).  What it buys: when both partners are
// line-aligned long jumps — most leaders on long paths — the two runs have the SAME a-side blocks, and a trip computes
// both terms of a lane from one load of its a-side record and position, with one add for the a-side (TWIN trip,
// sgd_kernels_1d.hip twin_trip): 3 blocks of records, position loads and atomic requests for 128 updates instead of 4.
// The kernel is bound by the memory side's atomic units, then by HBM bytes; this takes a quarter off both.
// A team wave works through an iteration in CHUNKS of this many updates (sgd_kernels_1d.hip, work pools); the rank cut-off
// that makes a count exact applies at the end of every chunk.  2048 = 32 full trips.
constexpr uint32_t TEAM_CHUNK = 2048;        // (the value of KArgs.chunk unless a probe says otherwise: capi.hip gfs_ctx_run_range)
// ... and of this many in the layout kernels: their pool is ONE counter per iteration (sgd_kernels_nd_team.hip K2c), and half as many
// claims are worth 3 % (C4: 49.5 -> 51.0 G updates/s; 8192: 50.8; profiles/r03/chunk_size_probe.log).  The sort is best at 2048.
constexpr uint32_t ND_TEAM_CHUNK = 4096;

// LONG RUNS.  A leader is expanded not over one trip but over K consecutive trips of its wave: trip `seg` takes the
// steps seg*B .. seg*B+B-1 further along the path, all with the leader's jump, so a run is K*B consecutive steps.
// Why: every term of a run moves its two nodes by about the same amount (same path, same interval: the same
// disagreement between this path's distance and the layout's), i.e. a run shifts two BLOCKS rigidly and leaves a
// step at the block's edges.  With B-step blocks those edges are dense, and nothing repairs them once the schedule
// has cooled (sgd.rs:456: in the cooling half only jumps of 1, 2 and near-uniform long jumps are drawn), which showed
// as +30 % relative error at path distances around B on large bubble graphs.  Longer blocks have proportionally fewer
// edges: with K*B = 1024 the error profile over all path distances is within a few % of the reference's independent
// terms, and below it at short distances (oracle mirror, profiles/r02/long_runs_mirror.log).  K adapts to the
// leader's path: the largest power of two <= a.chain with K*B <= cnt/4, so short paths keep short runs.  K depends
// on the path only, never on the jump: every leader of a path stands for the same number of terms, which keeps the
// distribution of jumps over the TERMS what the reference's sampler makes it.
__device__ __forceinline__ uint32_t run_trips(uint32_t chain, uint32_t bundle, uint32_t cnt) {
    const uint32_t room = cnt / (4u * bundle);
    if (room < 2u || chain < 2u) return 1u;
    const uint32_t p2 = 1u << (31 - __clz((int)room));
    return p2 < chain ? p2 : chain;
}
// Where trip `seg` of a run starts, in steps after the run's first step.  A long-jump run is contiguous (seg*B): that is
// the point of it.  A leader whose jump is shorter than a trip (|jump| < B) spreads its K trips evenly over the path
// (seg * cnt/K, wrapping): its terms are local — there is no block to shift — and contiguous short-jump runs have two
// drawbacks: the next trip reads nodes the previous one has just added to (no-return atomics are posted: the load can
// still see the old value), and two waves sweeping such runs along the same nodes at the same pace stay in each
// other's way for thousands of steps (path distance 1 lost precision as K grew, profiles/r02).
__device__ __forceinline__ uint32_t run_offset(uint32_t bundle, uint32_t cnt, uint32_t k, uint32_t ra0, uint32_t rb0, uint32_t seg) {
    const uint32_t z = ra0 < rb0 ? rb0 - ra0 : ra0 - rb0;
    return seg * (z < bundle ? cnt / k : bundle);
}

// One draw of the partner of step a0 (rank in its path) — sgd.rs:456-497 — and the line alignment of the two runs.
// ra = a0 on entry; on return the first steps of the a-run and of the b-run, and the partner's ok bits (Leader.ok).
__device__ __forceinline__ uint32_t draw_partner(const KArgs &a, const double *zeta_tab, Rng &rng, const uint32_t cnt,
                                                 const uint32_t thresh, const uint32_t slot0, uint32_t &ra, uint32_t &rb) {
    rb = ra;
    if (a.it.cooling || rng.flip() == 1u) {                                            // :456
        bool back = false, fwd = false;
        if (ra > 0u && (rng.flip() == 1u || ra == cnt - 1u)) back = true;              // :460
        else if (ra < cnt - 1u) fwd = true;                                            // :475
        if (back || fwd) {
            uint32_t room = back ? ra : (cnt - ra - 1u);
            uint32_t jump = a.space < room ? a.space : room;                           // :462,477
            double zeta = zeta_tab[space_index(a, jump)];
            uint32_t z = dirty_zipf(a.it, jump, zeta, rng.f64());                      // :472-473
            if (back) rb = ra >= z ? ra - z : 0u;                                      // :474
            else { uint64_t t = (uint64_t)ra + z; rb = t < cnt - 1u ? (uint32_t)t : cnt - 1u; }  // :489
        }
    } else {
        rb = rng.uniform32(cnt, thresh);                                               // :493-494
    }
    uint32_t ok = (rb != ra) ? 1u : 0u;                                                // :497
    if (ok && cnt >= 2u * a.bundle && slot0 != 0xFFFFFFFFu && !(a.dbg & 0x10u)) {        // only where a run will be expanded
        // Line-aligned runs, for jumps well beyond the run length.  Along a path laid out in slot order, a run that
        // starts on a multiple of 8 slots covers 8 lines of the position vector instead of 8.9.
        //  * The run starts sh = (slot of the leader's node) mod 8 steps before the leader (the leader's own term
        //    stays in the run, at lane sh).  (Bundles of 4 align to 4 slots: with 8 the leader could fall outside its
        //    own run and half of the nodes would never lead a long-range term.)
        //  * The partner run is a B-step block that starts zp = jump - r steps after it, r = jump mod 8, so it opens a
        //    line too, and lane l is paired with the block's step (l + r) mod B.  Lanes l < B - r keep EXACTLY the
        //    sampled jump; the last r <= 7 lanes wrap to the block's first steps, i.e. their jump is B steps shorter
        //    (or longer, for a backward jump) — about one long-range term in 18 moves by one run length.
        //  * Needs both blocks inside the path and apart (|zp| >= B); otherwise only the first rule is applied, and
        //    jumps shorter than B + 8 are left alone altogether (there only some lanes act — node-disjoint rule — and
        //    a fixed phase would leave some neighbour pairs never sampled).
        // C3: 0.30 -> 0.26 atomic requests per update, 68 -> 74 G updates/s.
        const int64_t jump = (int64_t)rb - (int64_t)ra;
        const int64_t Bn = (int64_t)a.bundle;
        const int64_t Rn = Bn * (int64_t)run_trips(a.chain, a.bundle, cnt);           // steps of the whole run (|jump| >= B + 8: contiguous)
        if (jump >= Bn + 8 || jump <= -(Bn + 8)) {
            const int64_t A = a.bundle < 8u ? (int64_t)a.bundle : 8;               // runs shorter than a line: align to the run length
            const uint32_t sh = slot0 & (uint32_t)(A - 1);
            if (ra >= sh) {
                const int64_t na = (int64_t)ra - (int64_t)sh;
                const int64_t r = ((jump % A) + A) % A, zp = jump - r, nb = na + zp;
                if (!(a.dbg & 0x20u) && na + Rn <= (int64_t)cnt && nb >= 0 && nb + Rn <= (int64_t)cnt && (zp >= Bn || zp <= -Bn)) {
                    ra = (uint32_t)na; rb = (uint32_t)nb; ok = 1u | 2u | ((uint32_t)r << 2);
                } else if (rb >= sh) { ra -= sh; rb -= sh; }
            }
        }
    }
    return ok;
}

template <bool LDS_TABLES>
__device__ __forceinline__ Leader sample_leader(const KArgs &a, const uint4 *path_tab, const double *zeta_tab, Rng &rng) {
    Leader L;
    const uint64_t s0 = sample_step(a, rng);                                           // sgd.rs:444
    const uint4 r0 = a.step_rec[s0];
    const uint4 pr = path_tab[rec_path(r0)];                                           // :445-446
    L.first_lo = pr.x; L.first_hi = pr.w; L.cnt = pr.y;
    L.ra0 = (uint32_t)(s0 - path_first(pr)); L.rb0 = L.ra0;                            // :452-453
    L.ra1 = L.ra0; L.rb1 = L.ra0;
    L.ok = 0;
    if (L.cnt == 1u) return L;                                                         // :448
    L.ok = draw_partner(a, zeta_tab, rng, L.cnt, pr.z, r0.x, L.ra0, L.rb0);
    if (a.partners == 2u) L.ok |= draw_partner(a, zeta_tab, rng, L.cnt, pr.z, r0.x, L.ra1, L.rb1) << 8;
    return L;
}

// A leader whose jump is shorter than the run (|jump| < B, on a path long enough to be expanded) is executed in TWO
// trips, one per COLOUR: the run's terms (a+l, a+l+z), l = 0..B-1, chain through shared nodes (a_{l+z} = b_l), so
// colour 0 holds the lanes with floor(l/|z|) even and colour 1 those with it odd — each colour is node-disjoint, and
// together the two trips apply every term of the run exactly once, like a long-jump run does in one trip.  (Round 1
// executed colour 0 only: short jumps then got half the updates the reference gives them, and the layout lost
// precision exactly at step distances below B — profiles/r02/quality_probe_before.log.)
template <int B>
__device__ __forceinline__ bool two_colour(uint32_t ok, uint32_t cnt, uint32_t ra0, uint32_t rb0) {
    if ((ok & 3u) != 1u || cnt < 2u * B) return false;
    const int64_t s = (int64_t)rb0 - (int64_t)ra0;
    return s < (int64_t)B && s > -(int64_t)B;                                          // s != 0 since ok & 1
}

// Expand leader values (already broadcast to this lane) into this lane's own term for the trip that starts `off` steps
// after the run's first step (run_offset) and colour `colour` (0 for every leader; 1 only for two_colour leaders).
// Returns false when the lane does not act in this trip.
template <int B>
__device__ __forceinline__ bool expand_run(uint32_t ok, uint64_t first, uint32_t cnt, uint32_t ra0, uint32_t rb0,
                                           int sub, uint32_t colour, uint32_t off, uint64_t &sa, uint64_t &sb) {
    if (!(ok & 1u)) return false;
    if (ok & 2u) {                                                                     // both runs line-aligned blocks inside the path
        if (colour) return false;
        sa = first + ra0 + off + (uint32_t)sub;
        sb = first + rb0 + off + (((uint32_t)sub + ((ok >> 2) & 7u)) & (uint32_t)(B - 1));
        return true;
    }
    uint32_t ra_l = ra0, rb_l = rb0;
    const uint32_t l = off + (uint32_t)sub;                                            // place in the run
    if (l != 0 || colour) {
        if (cnt < 2u * B) return false;                                                // short path: leader only
        const int64_t shift = (int64_t)rb0 - (int64_t)ra0;
        const uint32_t z = (uint32_t)(shift < 0 ? -shift : shift);
        if (z < (uint32_t)B) { if (((l / z) & 1u) != colour) return false; }           // node-disjoint lanes of this colour
        else if (colour) return false;
        uint64_t ra_w = (uint64_t)ra0 + l;                                             // wrap to the path start (l < cnt + B)
        if (ra_w >= cnt) ra_w -= cnt;
        if (ra_w >= cnt) ra_w -= cnt;
        ra_l = (uint32_t)ra_w;
        int64_t t = (int64_t)ra_l + shift;
        if (t < 0 || t > (int64_t)cnt - 1) {
            // partner outside the path.  A step this close to a path end samples into the path in
            // the reference (sgd.rs:460-490: the side without room is never chosen), so mirror the
            // jump; only for |jump| >= B, where mirrored and straight lanes cannot share a node.
            if (z < (uint32_t)B) return false;
            t = (int64_t)ra_l - shift;
            if (t < 0 || t > (int64_t)cnt - 1) return false;                           // never clamped
        }
        rb_l = (uint32_t)t;
    }
    sa = first + ra_l;                                                                 // :502
    sb = first + rb_l;                                                                 // :503
    return true;
}

__device__ __forceinline__ double rec_pos(const uint4 &r) {
    return (double)rec_pos_u64(r);
}

}  // namespace gfs
