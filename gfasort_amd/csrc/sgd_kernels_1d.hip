// sgd_kernels_1d.hip — K1 (reference streams) and K1b (team kernel) of path_linear_sgd, plus the
// multi-GPU replica-merge kernels.  See sgd_kernel_common.h / sgd_device.h.
#include "sgd_kernel_common.h"

namespace gfs {

// ------------------------------------------------------------------------------------------
// K1: 1D reference streams — one lane = one reference worker thread (sgd.rs:429-590)
// ------------------------------------------------------------------------------------------
// Everything of one loop trip that does not depend on the positions: the pair sampler and the rejections of
// sgd.rs:444-538.  Returns false where the reference `continue`s.
struct RefTerm1D { uint32_t i, j; int crowd; double term_dist; };

// (step_idx, ra: the trip's step a and its record, drawn and requested by the caller — see ref_run_1d)
template <bool LDS_TABLES>
__device__ __forceinline__ bool ref_sample_1d(const KArgs &a, const uint4 *path_tab, const double *zeta_tab, Rng &rng,
                                              const uint64_t step_idx, const uint4 &ra, RefTerm1D &t) {
    uint4 rb; uint64_t sa, sb; uint32_t cnt, path;
    if (!sample_pair_from<LDS_TABLES>(a, path_tab, zeta_tab, rng, step_idx, ra, rb, sa, sb, cnt, path)) return false;
    t.term_dist = fabs(rec_pos(ra) - rec_pos(rb));                                     // sgd.rs:513
    if (t.term_dist == 0.0) return false;                                              // :514
    t.crowd = crowd_shift<false>(a, ra, rb);
    t.i = ra.x; t.j = rb.x;
    return t.i != 0xFFFFFFFFu && t.j != 0xFFFFFFFFu;                                   // :525-538
}

// The worker loop for `quota` successful updates (sgd.rs:442-584).
// ONE thing is moved: the draw of the NEXT trip's step a (sgd.rs:444 — the next random number in the stream's order whatever
// happens in between) and the request of its record are issued BEFORE the current term's two adds instead of after them.
// vmcnt counts in order on gfx9: a load issued after the adds cannot be seen to complete before the adds have completed at the
// memory side (~1.5 us), and that wait was on every update's critical path; a load issued before them can.  Records are
// read-only, positions are still read after the previous term's adds: one stream is bit for bit the oracle's (tested), and a
// term is in flight no longer than before.  (Round 3 first overlapped the WHOLE next sample with the position loads: no faster,
// and a term's positions were then read ~1 us earlier — more terms in flight per stream, which the streams-per-node bound
// exists to limit: a tandem-repeat graph stable at the bound diverged.  profiles/r03/ref_fused_probe.log, repeat_stability.log.)
template <bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__device__ __forceinline__ void ref_run_1d(const KArgs &a, const uint4 *path_tab, const double *zeta_tab, Rng &rng,
                                           const uint32_t quota, const uint64_t max_att, const uint32_t tid,
                                           uint32_t &done, uint32_t &att, uint32_t &ntr) {
    double *x = a.x;
    uint32_t d = 0; uint64_t t = 0;
    uint64_t s_a = 0; uint4 r_a = make_uint4(0, 0, 0, 0); bool drawn = false;         // the next trip's step a, when drawn ahead
    while (d < quota && t < max_att) {
        ++t;
        if (!drawn) { s_a = sample_step(a, rng); r_a = a.step_rec[s_a]; }              // :444
        drawn = false;
        RefTerm1D cur;
        if (!ref_sample_1d<LDS_TABLES>(a, path_tab, zeta_tab, rng, s_a, r_a, cur)) continue;
        const double mu = crowd_scale(fmin(a.it.eta * (1.0 / cur.term_dist), 1.0), cur.crowd);   // :518-520
        double dx;
        if (a.dbg & 2u) dx = (double)cur.i - (double)cur.j;                            // ablation: no position loads
        else dx = load_pos<ATOMIC_LOADS>(x + cur.i) - load_pos<ATOMIC_LOADS>(x + cur.j);   // :541-543
        if (dx == 0.0) dx = 1e-9;                                                      // :546-548
        const double mag = fabs(dx);                                                   // :551
        const double delta = mu * (mag - cur.term_dist) / 2.0;                         // :552
        const double r = delta / mag;                                                  // :570
        const double r_x = r * dx;                                                     // :571
        if (d + 1u < quota && t < max_att) { s_a = sample_step(a, rng); r_a = a.step_rec[s_a]; drawn = true; }   // the next trip's :444
        if (a.dbg & 1u) { asm volatile("" :: "v"(r_x)); }                              // ablation: no atomics
        else {
            add_pos(x + cur.i, -r_x);                                                  // :575
            add_pos(x + cur.j, r_x);                                                   // :576
        }
        ++d;                                                                           // :579
        if (TRACE) {
            if (ntr < a.trace_per_stream) {
                TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ntr;
                tt->i = cur.i; tt->j = cur.j; tt->d = cur.term_dist;
                ++ntr;
            }
        }
    }
    done += d;
    att += t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
}

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
        const uint64_t max_att = max_att64 > 0xFFFFFFFFull ? 0xFFFFFFFFull : max_att64;
        uint32_t ntr = TRACE ? a.trace_cnt[tid] : 0;
        ref_run_1d<LDS_TABLES, ATOMIC_LOADS, TRACE>(a, path_tab, zeta_tab, rng, quota, max_att, tid, done, att, ntr);
        a.rng[tid] = rng.s0; a.rng[T + tid] = rng.s1; a.rng[2 * T + tid] = rng.s2; a.rng[3 * T + tid] = rng.s3;
        if (TRACE) a.trace_cnt[tid] = ntr;
    }
    flush_counters(a, done, att);
}

// K1d: the same streams, a range of iterations in one persistent launch with work pools (sgd_kernel_common.h
// ref_pooled_walk).  RNG state stays in registers for the whole schedule.
template <bool LDS_TABLES>
__global__ void sgd1d_fused_kernel(const KArgs a0, const IterConsts *its, const uint32_t n_iters, uint32_t *pool) {
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
        ref_run_1d<LDS_TABLES, true, false>(a, path_tab, zeta_tab, rng, share, max_att, tid, done, att, ntr);
    });
    if (live) { a.rng[tid] = rng.s0; a.rng[T + tid] = rng.s1; a.rng[2 * T + tid] = rng.s2; a.rng[3 * T + tid] = rng.s3; }
    flush_counters(a, done, att);
}

// ------------------------------------------------------------------------------------------
// K1b: 1D team kernel — bundled ("run") sampling (sgd_device.h).  A wave is a team:
//   pass   : all 64 lanes sample one leader term each from their own reference streams — the
//            Zipf/f64 arithmetic runs at full SIMD width instead of on one lane per bundle;
//   trips  : the leaders are executed one slot after the other, each as 64/B runs of B lanes.  With B = 64 a
//            leader's run extends over K consecutive trips (LONG RUNS, sgd_device.h run_trips); a leader whose
//            jump is shorter than a trip runs for one trip only and both of its colours (two_colour) are computed in
//            that trip (fused_trip), or in two trips where it touches a path end.  The records of the next
//            trip are requested before the current one is consumed, so a trip exposes one memory round trip
//            (its position loads); its adds are issued at once and never waited for by themselves.
//            (Round 1 issued a trip's adds one trip late, behind the next trip's loads; with long runs — whose
//            next trip touches the neighbouring lines — that was slower, 66.8 vs 73.4 G updates/s on C3, and it
//            let a wave read positions it was about to change: profiles/r02/quality_probe_defer.log.)
// The quota is per WAVE with a rank cut-off in the last trip: an iteration performs exactly its
// number of updates; what is left of a pass when the quota fills serves the next iteration (TeamState).
// ------------------------------------------------------------------------------------------
// Per-wave state that survives from one iteration to the next inside a launch.
struct TeamState {
    Rng rng;
    // the wave's current pass: 64 leaders (one per lane) of which `left` trip slots have not been fully expanded yet
    // (`colour` = 1: the current slot's first colour is done, its second is next — sgd_device.h two_colour).
    // A pass outlives the iteration it was sampled in (eta is not part of sampling); it is dropped when
    // the cooling phase — the only thing the sampler depends on besides the RNG — changes.
    Leader L = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t left = 0, cool = 0, colour = 0, seg = 0;   // seg: next trip of the current slot's run (sgd_device.h run_trips)
    uint32_t p = 0;                                     // partner of the current slot's leader the next trip belongs to
    uint32_t done = 0, att = 0, ntr = 0;
};

// One trip = (slot t of the pass, trip seg of its run, partner, colour): what every lane needs to execute it.
struct Trip {
    uint64_t sa = 0, sb = 0;
    uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
    uint4 rc = make_uint4(0, 0, 0, 0);   // twin trip: the record of the second partner's step
    bool twin = false;         // wave-uniform: both partners of an aligned leader in ONE trip (twin_trip)
    bool valid = false;        // this lane acts in the trip (fused trip: this lane's partner lies beyond the trip's 64 steps)
    int mshift = 0;            // != 0: merged short-jump trip (sgd_kernel_common.h merged_trip_shift)
    bool two = false;          // wave-uniform: some run of this slot has a second colour
    bool fused = false;        // wave-uniform: both colours of a short-jump run in ONE trip (fused_trip)
    uint32_t k = 1;            // wave-uniform: trips of this slot's run (long runs: B = 64 only)
    uint32_t off = 0;          // wave-uniform: this trip starts `off` steps after the run's first step (run_offset)
};

template <int B>
__device__ __forceinline__ void expand_trip(const KArgs &a, const Leader &L, int t, uint32_t seg, uint32_t p, uint32_t colour, int sub, int q, Trip &tr) {
    constexpr int RUNS = 64 / B;
    const int ll = t * RUNS + q;
    const uint32_t okw = bcast<B>(L.ok, ll), cnt = bcast<B>(L.cnt, ll);
    const uint32_t ok = leader_ok(okw, p);
    const uint32_t ra0 = p ? bcast<B>(L.ra1, ll) : bcast<B>(L.ra0, ll), rb0 = p ? bcast<B>(L.rb1, ll) : bcast<B>(L.rb0, ll);
    const uint64_t first = bcast_first<B>(L, ll);
    // long runs only where the whole wave follows one leader; a leader the reference rejected takes one (empty) trip.
    // (The number of trips must not depend on the partner: the trips of a slot go seg by seg, both partners each.)
    tr.k = (B == 64 && ((okw | (okw >> 8)) & 1u) && cnt >= 2u * B) ? run_trips(a.chain, (uint32_t)B, cnt) : 1u;
    tr.off = B == 64 ? run_offset((uint32_t)B, cnt, tr.k, ra0, rb0, seg) : 0u;
    tr.mshift = merged_trip_shift<B>(ok, cnt, ra0, rb0, tr.off);
    const bool two = !(a.dbg & 0x08u) && two_colour<B>(ok, cnt, ra0, rb0);
    tr.two = B == 64 ? two : (__any(two) != 0);                       // B = 64: the leader is wave-uniform already
    tr.ra = make_uint4(0, 0, 0, 0); tr.rb = make_uint4(0, 0, 0, 0); tr.rc = make_uint4(0, 0, 0, 0);
    tr.fused = B == 64 && tr.mshift != 0 && colour == 0 && two && !(a.dbg & 0x100u);
    // both partners line-aligned long jumps: their a-runs are the same blocks (draw_partner), one trip serves both
    tr.twin = B == 64 && p == 0u && a.partners == 2u && (okw & 3u) == 3u && ((okw >> 8) & 3u) == 3u && !(a.dbg & 0x04u);
    const uint32_t rb1 = bcast<B>(L.rb1, ll);
    if (tr.twin) {
        // ... unless a block of one partner run lies within two trips of the other's (|gap| < 192 steps): the wave would read, as
        // one partner's positions, what it has only just added as the other's — in the same trip from the very snapshot the
        // add was computed from (no-return atomics are posted).  Partner runs further apart may overlap as runs: their common
        // nodes are then read two or more trips after they were added to.  (With round 1's free-running launch such leaders as
        // twin trips cost the 525k-node bubble graph a quarter of its precision at path distance 1, profiles/r02/two_partners.log.)
        const int64_t gap = (int64_t)rb0 - (int64_t)rb1, lim = 192;
        if (gap < lim && gap > -lim) tr.twin = false;
    }
    if (tr.twin) {
        tr.sa = first + ra0 + tr.off + (uint32_t)sub;
        tr.sb = first + rb0 + tr.off + (((uint32_t)sub + ((okw >> 2) & 7u)) & 63u);
        const uint64_t sc = first + rb1 + tr.off + (((uint32_t)sub + ((okw >> 10) & 7u)) & 63u);
        tr.valid = true;
        tr.ra = a.step_rec[tr.sa]; tr.rb = a.step_rec[tr.sb]; tr.rc = a.step_rec[sc];
        return;
    }
    if (tr.fused) {
        // every lane takes its own step of the trip and its partner's record (the partners inside the trip are the other
        // lanes' own steps: the same lines, no extra traffic; merged_trip_shift guarantees all of them lie in the path)
        const int dst = sub + tr.mshift;
        tr.sa = first + merged_trip_base(cnt, ra0, tr.off) + (uint32_t)sub;
        tr.sb = (uint64_t)((int64_t)tr.sa + tr.mshift);
        tr.valid = dst < 0 || dst > 63;
        tr.ra = a.step_rec[tr.sa];
        tr.rb = a.step_rec[tr.sb];
        return;
    }
    tr.valid = expand_run<B>(ok, first, cnt, ra0, rb0, sub, colour, tr.off, tr.sa, tr.sb);
    if (tr.valid) { tr.ra = a.step_rec[tr.sa]; tr.rb = a.step_rec[tr.sb]; }
}

// The term arithmetic of sgd.rs:518-571 on values already in registers; returns r_x.
__device__ __forceinline__ double term_move(const KArgs &a, double term_dist, double xi, double xj, int crowd) {
    double mu = crowd_scale(fmin(a.it.eta * (1.0 / term_dist), 1.0), crowd);          // :518-520
    double dx = xi - xj;                                                               // :543
    if (dx == 0.0) dx = 1e-9;                                                          // :546-548
    double mag = fabs(dx);                                                             // :551
    double delta = mu * (mag - term_dist) / 2.0;                                       // :552
    double r = delta / mag;                                                            // :570
    return r * dx;                                                                     // :571
}

__device__ __forceinline__ double shfl_f64(double v, int src) { return __shfl(v, src, 64); }

// FUSED short-jump trip (B = 64, |jump| = z < 64, the run's 64 steps and all their partners inside the path).
// The run's 64 terms (l, l+s) form two node-disjoint colours (two_colour).  As two trips, each colour loads the records
// and positions of half the lanes and of their partners — which are the OTHER colour's lanes — and the second colour
// reads what the first one wrote.  Here every lane loads its own record and position once; partners inside the run are
// read from the lane that holds them (wave shuffles), colour 1 computes on the positions colour 0 has just produced in
// registers (bit for bit what it would read back from memory when no other wave interferes), and a lane's node takes ONE
// add for both colours — what it gave as the acting lane of one colour plus what it took as the partner in the other:
// memory receives x + (-r + r') where two trips would make it (x - r) + r' (the oracle's mirror rounds the same way).
// Same terms, same arithmetic, same order as the two trips.  It buys precision under concurrency — both colours see ONE
// snapshot of the run's 64 nodes and land in one memory round trip, instead of exposing the run to the other ~5 000 waves
// for two: at one stream per two nodes (525k-node bubble graph) the relative error at path distance 1 is 0.200 against
// 0.248 with two trips (reference streams: 0.194; profiles/r02/quality_probe_long_runs.log) — and, since the one add per node, speed: the kernel is bound by the memory side's atomic units,
// and a fused trip now costs 8 requests for 64 updates where two trips cost 16 (C3 80.0 -> 87.4 G updates/s, bubble
// graphs 49.9 -> 53.6 and 52.0 -> 55.8: profiles/r02/fused_trip_one_add.log).  Returns false when the wave's quota filled
// before the second colour: the caller leaves that colour to the next iteration as a generic trip.
template <bool ATOMIC_LOADS, bool TRACE>
__device__ __forceinline__ bool fused_trip(const KArgs &a, TeamState &ts, const Trip &cur, const int lane,
                                           const uint32_t tid, const uint64_t wave_quota, uint64_t &wave_done) {
    double *x = a.x;
    const int s = cur.mshift, z = s < 0 ? -s : s;
    const int dst = lane + s, src = lane - s;                          // my partner's lane; the lane whose partner I am
    const bool out = cur.valid;                                        // partner beyond the trip's 64 steps
    const int dstc = out ? lane : dst, srcc = (src < 0 || src > 63) ? lane : src;
    const uint32_t grp = ((cur.off + (uint32_t)lane) / (uint32_t)z) & 1u;
    const uint32_t node = cur.ra.x, pnode = cur.rb.x;
    double xo = 0.0, xp = 0.0;                                         // my position; my partner's when it is beyond the trip
    if (!(a.dbg & 2u)) {
        if (node != 0xFFFFFFFFu) xo = load_pos<ATOMIC_LOADS>(x + node);
        if (out && pnode != 0xFFFFFFFFu) xp = load_pos<ATOMIC_LOADS>(x + pnode);       // (partners inside: from their lanes)
    } else { xo = (double)node; xp = (double)pnode; }
    const double term_dist = fabs(rec_pos(cur.ra) - rec_pos(cur.rb));                  // sgd.rs:513
    const int crowd = crowd_shift<true>(a, cur.ra, cur.rb);
    const bool term_ok = term_dist != 0.0 && node != 0xFFFFFFFFu && pnode != 0xFFFFFFFFu;   // :514, :525-538
    double acc = 0.0;
    bool touched = false, second = true;
#pragma unroll
    for (uint32_t colour = 0; colour < 2u; ++colour) {
        ++ts.att;
        bool valid = term_ok && grp == colour;
        const unsigned long long vmask = __ballot(valid);
        const uint64_t remaining = wave_quota - wave_done;
        const uint32_t nvalid = (uint32_t)__popcll(vmask);
        if (valid && nvalid > remaining) valid = (uint32_t)__popcll(vmask & ((1ull << lane) - 1ull)) < remaining;
        wave_done += nvalid < remaining ? nvalid : remaining;
        const double xpart_in = shfl_f64(xo, dstc);                    // partner's CURRENT position (colour 0's result for colour 1)
        const double xj = out ? xp : xpart_in;
        double r_x = 0.0;
        if (valid) {
            r_x = term_move(a, term_dist, xo, xj, crowd);
            ++ts.done;                                                                 // :579
            if (TRACE) {
                if (ts.ntr < a.trace_per_stream) {
                    TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ts.ntr;
                    tt->i = node; tt->j = pnode; tt->d = term_dist;
                    ++ts.ntr;
                }
            }
        }
        // the +r of the lane whose partner I am
        // (every shuffle is a statement of its own: inside `a && __shfl(..)` or `c ? x : __shfl(..)` the compiler may run it
        // only on the lanes that need the result, and a SOURCE lane that is switched off then supplies nothing)
        const double rv = shfl_f64(r_x, srcc);
        const int vsrc = __shfl((int)valid, srcc, 64);
        const bool recv = src >= 0 && src <= 63 && vsrc != 0;
        // a lane acts or receives in a colour, never both (its group's parity decides)
        if (valid) xo = xo - r_x;                                                      // :575  x[i] - r_x
        if (recv) xo = xo + rv;                                                        // :576  x[j] + r_x
        // a lane's own node takes ONE add for both colours, the sum of what it gave as an acting lane in one colour and
        // took as a partner in the other (its register holds (x - r) + r', memory receives x + (-r + r')): half the
        // atomic requests of the trip
        if (valid) { acc = touched ? acc - r_x : -r_x; touched = true; }
        if (recv) { acc = touched ? acc + rv : rv; touched = true; }
        if (valid && out && !(a.dbg & 1u)) add_pos(x + pnode, r_x);                    // partner beyond the trip
        if (colour == 0 && wave_done >= wave_quota) { second = false; break; }
    }
    if (touched && !(a.dbg & 1u)) add_pos(x + node, acc);
    return second;
}

// TWIN trip (B = 64, two partners, both line-aligned long jumps: sgd_device.h Leader).  A lane's step a is the a-side of two
// terms, (a, b) and (a, c), b and c in two other aligned blocks of the path.  One load of a's record and position serves
// both; the second term computes on what the first left in the register, as it would read it back from memory; a's node
// takes ONE add, -(r + r'), b and c one each: 3 blocks of 8 requests for 128 updates where two trips take 4.  Returns false
// when the wave's quota filled before the second term: the caller leaves the second partner's trip to the next iteration.
template <bool ATOMIC_LOADS, bool TRACE>
__device__ __forceinline__ bool twin_trip(const KArgs &a, TeamState &ts, const Trip &cur, const int lane,
                                          const uint32_t tid, const uint64_t wave_quota, uint64_t &wave_done) {
    double *x = a.x;
    const uint32_t node = cur.ra.x, nb = cur.rb.x, nc = cur.rc.x;
    double xa = 0.0, xb = 0.0, xc = 0.0;
    if (!(a.dbg & 2u)) {
        if (node != 0xFFFFFFFFu) xa = load_pos<ATOMIC_LOADS>(x + node);
        if (nb != 0xFFFFFFFFu) xb = load_pos<ATOMIC_LOADS>(x + nb);
        if (nc != 0xFFFFFFFFu) xc = load_pos<ATOMIC_LOADS>(x + nc);
    } else { xa = (double)node; xb = (double)nb; xc = (double)nc; }
    const double pa = rec_pos(cur.ra);
    double acc = 0.0;
    bool touched = false, second = true;
#pragma unroll
    for (uint32_t p = 0; p < 2u; ++p) {
        ++ts.att;
        const uint4 &rp = p ? cur.rc : cur.rb;
        const uint32_t pn = p ? nc : nb;
        const double term_dist = fabs(pa - rec_pos(rp));                               // sgd.rs:513
        bool valid = term_dist != 0.0 && node != 0xFFFFFFFFu && pn != 0xFFFFFFFFu;     // :514, :525-538
        const unsigned long long vmask = __ballot(valid);
        const uint64_t remaining = wave_quota - wave_done;
        const uint32_t nvalid = (uint32_t)__popcll(vmask);
        if (valid && nvalid > remaining) valid = (uint32_t)__popcll(vmask & ((1ull << lane) - 1ull)) < remaining;
        wave_done += nvalid < remaining ? nvalid : remaining;
        if (valid) {
            const double r_x = term_move(a, term_dist, xa, p ? xc : xb, crowd_shift<true>(a, cur.ra, rp));   // :518-571
            ++ts.done;                                                                 // :579
            if (TRACE) {
                if (ts.ntr < a.trace_per_stream) {
                    TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ts.ntr;
                    tt->i = node; tt->j = pn; tt->d = term_dist;
                    ++ts.ntr;
                }
            }
            xa = xa - r_x;                                                             // :575
            acc = touched ? acc - r_x : -r_x; touched = true;
            if (!(a.dbg & 1u)) add_pos(x + pn, r_x);                                   // :576
        }
        if (p == 0u && wave_done >= wave_quota) { second = false; break; }
    }
    if (touched && !(a.dbg & 1u)) add_pos(x + node, acc);
    return second;
}

// One SGD iteration of one wave: passes and trips until the wave's quota is filled.
template <int B, bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__device__ __forceinline__ void team_iteration(const KArgs &a, const uint4 *path_tab, const double *zeta_tab,
                                               TeamState &ts, const uint32_t tid, const uint64_t wave_quota, const IterConsts *itp = nullptr) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & (B - 1);
    const int q = lane / B;
    const uint64_t max_passes = (uint64_t)a.attempt_factor * (wave_quota / (64u * B) + 1u) + 16u;
    uint64_t wave_done = 0, passes = 0;
    double *x = a.x;
    while (wave_done < wave_quota && passes < max_passes) {
        if (ts.left == 0 || ts.cool != (uint32_t)a.it.cooling) {
            ++passes;
            // (the sampler reads its launch constants afresh — sgd_kernel_common.h reload_kargs — and, in a fused launch, the
            // iteration's constants from the schedule in memory: itp)
            KArgs as;
            reload_kargs(as);
            if (itp) {
                const IterConsts *ip = itp;
                asm volatile("" : "+s"(ip));
                as.it = *ip;
            }
            ts.L = sample_leader<LDS_TABLES>(as, path_tab, zeta_tab, ts.rng);
            ts.left = B; ts.cool = (uint32_t)a.it.cooling; ts.colour = 0; ts.seg = 0; ts.p = 0;
        }
        const Leader &L = ts.L;
        int t = B - (int)ts.left;
        uint32_t colour = ts.colour, seg = ts.seg, p = ts.p;
        Trip cur;
        expand_trip<B>(a, L, t, seg, p, colour, sub, q, cur);          // expand and request the records of the first trip
        for (;;) {
            // the trip after this one: this trip's second colour (unless fused into it), else the same trip of the run for
            // the leader's second partner (unless this is a twin trip), else the run's next trip, else the next slot;
            // request its records now
            int t_n = t; uint32_t colour_n = 0u, seg_n = seg, p_n = p;
            if (colour == 0 && cur.two && !cur.fused) colour_n = 1u;
            else if (p == 0u && a.partners == 2u && !cur.twin) p_n = 1u;
            else if (seg + 1u < cur.k) { seg_n = seg + 1u; p_n = 0u; }
            else { t_n = t + 1; seg_n = 0u; p_n = 0u; }
            const bool have_n = t_n < B;
            Trip nxt;
            if (have_n) expand_trip<B>(a, L, t_n, seg_n, p_n, colour_n, sub, q, nxt);
            if (B == 64 && cur.twin) {
                if (!twin_trip<ATOMIC_LOADS, TRACE>(a, ts, cur, lane, tid, wave_quota, wave_done)) {
                    ts.colour = 0u; ts.seg = seg; ts.p = 1u;           // quota filled between the partners: the second one
                    break;                                             // is the next iteration's first trip (generic form)
                }
                ts.colour = 0u; ts.seg = seg_n; ts.p = 0u;
                if (t_n != t) --ts.left;
                if (wave_done >= wave_quota || !have_n) break;
                cur = nxt; t = t_n; colour = colour_n; seg = seg_n; p = p_n;
                continue;
            }
            if (B == 64 && cur.fused) {
                if (!fused_trip<ATOMIC_LOADS, TRACE>(a, ts, cur, lane, tid, wave_quota, wave_done)) {
                    ts.colour = 1u; ts.seg = seg; ts.p = p;            // quota filled between the colours: the second one is
                    break;                                             // the next iteration's first trip (generic form)
                }
                ts.colour = 0u; ts.seg = seg_n; ts.p = p_n;
                if (t_n != t) --ts.left;
                if (wave_done >= wave_quota || !have_n) break;
                cur = nxt; t = t_n; colour = colour_n; seg = seg_n; p = p_n;
                continue;
            }
            // consume the current trip
            ++ts.att;
            ts.colour = colour_n; ts.seg = seg_n; ts.p = p_n;
            if (t_n != t) --ts.left;
            bool valid = cur.valid;
            const uint4 ra = cur.ra, rb = cur.rb;
            const int mshift = cur.mshift;
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
            if (valid) {
                if (a.dbg & 2u) { xi = (double)i; xj = (double)j; }                    // ablation: no position loads
                else { xi = load_pos<ATOMIC_LOADS>(x + i); xj = load_pos<ATOMIC_LOADS>(x + j); }   // :541-542
            }
            double r_x = 0.0;
            if (valid) {
                r_x = term_move(a, term_dist, xi, xj, crowd_shift<true>(a, ra, rb));   // :518-571
                ++ts.done;                                                             // :579
                if (TRACE) {
                    if (ts.ntr < a.trace_per_stream) {
                        TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ts.ntr;
                        tt->i = i; tt->j = j; tt->d = term_dist;
                        ++ts.ntr;
                    }
                }
            }
            // the adds of this trip (:575-576): -r_x to node i, +r_x to node j
            bool o1f = valid, o2f = valid; uint32_t o1s = i, o2s = j; double o1v = -r_x, o2v = r_x;
            if (B == 64 && mshift != 0) {
                // merged short-jump trip (wave-uniform branch): a resting lane takes over the +r of the acting lane
                // whose partner is the step it sits on
                const int z = mshift < 0 ? -mshift : mshift;
                const int src = lane - mshift;                                         // the lane whose partner I sit on
                const int srcc = src < 0 ? 0 : (src > 63 ? 63 : src);
                const double rv = __shfl(r_x, srcc, 64);
                const uint32_t js = (uint32_t)__shfl((int)j, srcc, 64);
                const int vs = __shfl((int)valid, srcc, 64);
                const bool resting = (((cur.off + (uint32_t)lane) / (uint32_t)z) & 1u) != colour;
                if (resting && src >= 0 && src < 64 && vs) { o1f = true; o1s = js; o1v = rv; }
                const int dst = lane + mshift;                                         // where my own partner sits
                o2f = valid && (dst < 0 || dst > 63);                                  // beyond the run: add it myself
            }
            if (!(a.dbg & 1u)) {                                                       // (ablation: no atomics)
                if (o1f) add_pos(x + o1s, o1v);
                if (o2f) add_pos(x + o2s, o2v);
            }
            if (wave_done >= wave_quota) break;                                        // what is left of the pass serves the next iteration
            if (!have_n) break;
            cur = nxt; t = t_n; colour = colour_n; seg = seg_n; p = p_n;
        }
    }
}

// lead word: ok bits of partner 0 (0..7) | trips left (8..15) | cooling (16) | colour (17) | seg (18..25) | partner (26) |
// ok bits of partner 1 (27..31)
__device__ __forceinline__ void load_pass(const KArgs &a, uint32_t tid, TeamState &ts) {
    if (!a.lead) return;
    const uint64_t T = a.n_streams;
    ts.L.first_lo = a.lead[tid]; ts.L.first_hi = a.lead[T + tid]; ts.L.cnt = a.lead[2 * T + tid];
    ts.L.ra0 = a.lead[3 * T + tid]; ts.L.rb0 = a.lead[4 * T + tid];
    const uint32_t w = a.lead[5 * T + tid];
    ts.L.ra1 = a.lead[6 * T + tid]; ts.L.rb1 = a.lead[7 * T + tid];
    ts.L.ok = (w & 0xFFu) | ((w >> 27) << 8);
    // (the place in the pass is the same for all 64 lanes of the wave: scalar registers)
    const uint32_t ws = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
    ts.left = (ws >> 8) & 0xFFu; ts.cool = (ws >> 16) & 1u; ts.colour = (ws >> 17) & 1u; ts.seg = (ws >> 18) & 0xFFu; ts.p = (ws >> 26) & 1u;
}
__device__ __forceinline__ void store_pass(const KArgs &a, uint32_t tid, const TeamState &ts) {
    if (!a.lead) return;
    const uint64_t T = a.n_streams;
    a.lead[tid] = ts.L.first_lo; a.lead[T + tid] = ts.L.first_hi; a.lead[2 * T + tid] = ts.L.cnt;
    a.lead[3 * T + tid] = ts.L.ra0; a.lead[4 * T + tid] = ts.L.rb0;
    a.lead[5 * T + tid] = (ts.L.ok & 0xFFu) | (ts.left << 8) | (ts.cool << 16) | (ts.colour << 17) | (ts.seg << 18) | (ts.p << 26) |
                          (((ts.L.ok >> 8) & 0x1Fu) << 27);
    a.lead[6 * T + tid] = ts.L.ra1; a.lead[7 * T + tid] = ts.L.rb1;
}

__device__ __forceinline__ uint64_t wave_quota_of(const KArgs &a, uint32_t tid) {
    // wave quota = sum of its 64 lanes' per-stream quotas
    const uint32_t wave_first = tid & ~63u;
    uint64_t wq = (uint64_t)a.quota_base * 64u;
    if (wave_first < a.quota_rem) wq += (a.quota_rem - wave_first) < 64u ? (a.quota_rem - wave_first) : 64u;
    return wq;
}

// (4 waves per SIMD = 128 VGPRs: a twin trip keeps three blocks in flight and the next trip's records are on their way; built
// for 5 waves — 96 VGPRs — the kernel spills 58 registers and is slower: 88.5 against 91.5 G updates/s on C3 with round 1's
// launch, profiles/r02/two_partners.log.  Before the twin trips a fifth wave was worth +3 %.)
template <int B, bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__global__ void __attribute__((amdgpu_waves_per_eu(4, 4))) sgd1d_team_kernel(const KArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint4 *path_tab; const double *zeta_tab;
    stage_tables<LDS_TABLES>(a, smem, path_tab, zeta_tab);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;       // n_streams % 64 == 0 (host-checked)
    if (tid >= a.n_streams) return;                                   // whole waves only
    const uint64_t T = a.n_streams;
    TeamState ts;
    ts.rng.s0 = a.rng[tid]; ts.rng.s1 = a.rng[T + tid]; ts.rng.s2 = a.rng[2 * T + tid]; ts.rng.s3 = a.rng[3 * T + tid];
    ts.ntr = TRACE ? a.trace_cnt[tid] : 0;
    load_pass(a, tid, ts);
    const uint64_t wq = wave_quota_of(a, tid);                        // worked through in chunks, like a pool (K1c)
    for (uint64_t done = 0; done < wq; done += a.chunk)
        team_iteration<B, LDS_TABLES, ATOMIC_LOADS, TRACE>(a, path_tab, zeta_tab, ts, tid, wq - done < a.chunk ? wq - done : a.chunk);
    a.rng[tid] = ts.rng.s0; a.rng[T + tid] = ts.rng.s1; a.rng[2 * T + tid] = ts.rng.s2; a.rng[3 * T + tid] = ts.rng.s3;
    if (TRACE) a.trace_cnt[tid] = ts.ntr;
    store_pass(a, tid, ts);
    flush_counters(a, ts.done, ts.att);
}

// K1c: the same, FUSED over a range of iterations (single-GPU runs): one persistent launch in which every wave walks
// the schedule its[0..n_iters).  Saves the per-launch ramp, tail and RNG round trip.
//
// WORK POOLS.  Round 1 gave every wave a fixed quota per iteration and no grid barrier.  Free-running waves drift apart:
// one whose trips happen to be cheap runs iterations ahead of one whose trips are dear, so terms of several iterations —
// several values of eta — are applied side by side, and the last, finest iterations are finished by the stragglers alone.
// (The reference's iterations overlap by what its workers do in 1 ms, sgd.rs:366-403: a few per cent of an iteration.)
// Measured on the 525k-node bubble graph: relative error at path distance 1 of 0.195-0.246 depending on the stream count
// with free-running waves, 0.187-0.191 at every count with one launch per iteration — which is what the oracle's
// sequential mirror gives (profiles/r02/pacing.log).  A counting barrier per iteration (with a lag of 1-6 iterations)
// restores the precision but leaves the fast waves idle: C3 66-88 G updates/s against 93.
// Instead an iteration's min_term_updates updates are a POOL that the waves draw from in chunks of TEAM_CHUNK updates
// (one returning atomic per chunk, on one of up to 16 counters — one per 16 waves, sgd_kernel_common.h pool_slots — so that the
// claims do not queue on one address; a wave
// claims its next chunk before it works on the current one).  A wave moves on to iteration k + 1 when its counter of
// iteration k is exhausted: no wave is ever more than two chunks away from the others OF ITS COUNTER (the counters are fixed
// shares of an iteration: the waves of a fast one can run ahead of a slow one's — harmless for the sort, whose figures are the
// same with a launch per iteration; the layout kernel, K2c, uses one counter), nobody waits, and a wave that is
// slow simply takes fewer chunks — which is the reference's own rule (its workers share one count per iteration).  Every
// iteration still applies exactly min_term_updates updates with its own eta/theta.  C3: 97.8 G updates/s.
// (A single wave claims every chunk itself, in order: the kernel with fixed quotas works through its quota in the same
// chunks, so that one wave is bit for bit the oracle's mirror in both.)
// (POOL is a template parameter so that each build holds ONE inlined copy of the trip machine: with both launch modes in one
// kernel the pooled path spilled 65 VGPRs into 188 B of scratch per lane — and a first dispatch that needs more scratch than any
// kernel before it makes the runtime re-size the queue's scratch, the ~0.12 ms "first-dispatch latency" of profiles/r02/launch_gap.log.)
template <int B, bool LDS_TABLES, bool POOL>
__global__ void __attribute__((amdgpu_waves_per_eu(4, 4))) sgd1d_team_fused_kernel(const KArgs a0, const IterConsts *its, const uint32_t n_iters,
                                                                                   uint32_t *pool) {
    constexpr bool ATOMIC_LOADS = true;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint4 *path_tab; const double *zeta_tab;
    stage_tables<LDS_TABLES>(a0, smem, path_tab, zeta_tab);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= a0.n_streams) return;
    const uint64_t T = a0.n_streams;
    KArgs a = a0;
    TeamState ts;
    ts.rng.s0 = a.rng[tid]; ts.rng.s1 = a.rng[T + tid]; ts.rng.s2 = a.rng[2 * T + tid]; ts.rng.s3 = a.rng[3 * T + tid];
    const int lane = threadIdx.x & 63;
    load_pass(a, tid, ts);
    if (POOL) {
        const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), n_waves = a0.n_streams >> 6;   // (scalar registers)
        const uint32_t slots = pool_slots(n_waves), slot = wave % slots;
        const uint64_t total = (uint64_t)a0.quota_base * a0.n_streams + a0.quota_rem;
        const uint32_t cap = (uint32_t)(total / slots + (slot < total % slots ? 1u : 0u));   // < 2^31 (host-checked)
        uint32_t k = 0, claim = 0;
        a.it = its[0];
        if (lane == 0) claim = __hip_atomic_fetch_add(pool + slot * POOL_STRIDE, a0.chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (k < n_iters) {
            const uint32_t old = (uint32_t)__builtin_amdgcn_readfirstlane((int)claim);
            if (old >= cap) {                                          // this iteration's pool is exhausted
                if (++k == n_iters) break;
                a.it = its[k];                                         // wave-uniform: scalar loads
                if (lane == 0) claim = __hip_atomic_fetch_add(pool + ((size_t)k * POOL_SLOTS + slot) * POOL_STRIDE, a0.chunk,
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            // the next claim travels while this chunk is worked on
            if (lane == 0) claim = __hip_atomic_fetch_add(pool + ((size_t)k * POOL_SLOTS + slot) * POOL_STRIDE, a0.chunk,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            team_iteration<B, LDS_TABLES, ATOMIC_LOADS, false>(a, path_tab, zeta_tab, ts, tid, cap - old < a0.chunk ? cap - old : a0.chunk, its + k);
        }
    } else {
        // fixed quota per wave and iteration, free-running (GFS_F_DBG_FREE_RUNNING)
        const uint64_t wq = wave_quota_of(a, tid);
        for (uint32_t k = 0; k < n_iters; ++k) {
            a.it = its[k];
            for (uint64_t done = 0; done < wq; done += a.chunk)
                team_iteration<B, LDS_TABLES, ATOMIC_LOADS, false>(a, path_tab, zeta_tab, ts, tid, wq - done < a.chunk ? wq - done : a.chunk, its + k);
        }
    }
    a.rng[tid] = ts.rng.s0; a.rng[T + tid] = ts.rng.s1; a.rng[2 * T + tid] = ts.rng.s2; a.rng[3 * T + tid] = ts.rng.s3;
    store_pass(a, tid, ts);
    flush_counters(a, ts.done, ts.att);
}

// ------------------------------------------------------------------------------------------
// Multi-GPU replica merge (no reference equivalent; gfasort_amd/distributed.py).  Two streaming
// kernels around the one all-reduce of an iteration:
//   prepare: buf[0][k] = (float)(x[k] - x_prev[k])  (this rank's batch),  buf[1][k] = delta != 0
//   apply  : x_prev[k] += sum_delta[k] / max(1, sum_touched[k]);  x[k] = x_prev[k]
// The exchanged buffer is f32 (half the xGMI bytes; a delta is rounded to 24 bits, 6e-8 relative,
// far below the SGD noise; all ranks apply the same reduced values, so replicas stay identical).
// Grid-stride, pure HBM streaming.
// ------------------------------------------------------------------------------------------
__global__ void merge_prepare_kernel(const double *x, const double *x_prev, float *buf, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 2;
    for (uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; k < n; k += stride) {
        if (k + 1 < n) {
            const double2 a = *reinterpret_cast<const double2 *>(x + k), b = *reinterpret_cast<const double2 *>(x_prev + k);
            const float d0 = (float)(a.x - b.x), d1 = (float)(a.y - b.y);
            *reinterpret_cast<float2 *>(buf + k) = make_float2(d0, d1);
            // the touched row starts at buf + n: 8-byte aligned only when n is even
            if (n & 1) { buf[n + k] = d0 != 0.f ? 1.f : 0.f; buf[n + k + 1] = d1 != 0.f ? 1.f : 0.f; }
            else *reinterpret_cast<float2 *>(buf + n + k) = make_float2(d0 != 0.f ? 1.f : 0.f, d1 != 0.f ? 1.f : 0.f);
        } else {
            const float d = (float)(x[k] - x_prev[k]);
            buf[k] = d; buf[n + k] = d != 0.f ? 1.f : 0.f;
        }
    }
}
__global__ void merge_apply_kernel(double *x, double *x_prev, const float *buf, uint64_t n, double scale_all) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const double c = (double)buf[n + k];
        const double div = scale_all > 0.0 ? scale_all : (c > 1.0 ? c : 1.0);
        const double v = x_prev[k] + (double)buf[k] / div;
        x_prev[k] = v; x[k] = v;
    }
}
hipError_t launch_merge_prepare(const double *x, const double *x_prev, float *buf, uint64_t n, hipStream_t st) {
    hipLaunchKernelGGL(merge_prepare_kernel, dim3(2048), dim3(256), 0, st, x, x_prev, buf, n);
    return hipGetLastError();
}
hipError_t launch_merge_apply(double *x, double *x_prev, const float *buf, uint64_t n, double scale_all, hipStream_t st) {
    hipLaunchKernelGGL(merge_apply_kernel, dim3(2048), dim3(256), 0, st, x, x_prev, buf, n, scale_all);
    return hipGetLastError();
}

template <bool L, bool A>
static hipError_t launch_1d_t(const KArgs &a, bool trace, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (trace) hipLaunchKernelGGL((sgd1d_kernel<L, A, true>), grid, block, lds, st, a);
    else       hipLaunchKernelGGL((sgd1d_kernel<L, A, false>), grid, block, lds, st, a);
    return hipGetLastError();
}
template <int B, bool L, bool A>
static hipError_t launch_1db_t(const KArgs &a, bool trace, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (trace) hipLaunchKernelGGL((sgd1d_team_kernel<B, L, true, true>), grid, block, lds, st, a);   // debug trace: agent-scope loads
    else       hipLaunchKernelGGL((sgd1d_team_kernel<B, L, A, false>), grid, block, lds, st, a);
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
template <int B>
static hipError_t launch_1d_fused_b(const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables,
                                    uint32_t *pool, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (pool) {
        if (lds_tables) hipLaunchKernelGGL((sgd1d_team_fused_kernel<B, true, true>), grid, block, lds, st, a, d_its, n_iters, pool);
        else            hipLaunchKernelGGL((sgd1d_team_fused_kernel<B, false, true>), grid, block, 0, st, a, d_its, n_iters, pool);
    } else {                                                           // GFS_F_DBG_FREE_RUNNING: fixed quotas
        if (lds_tables) hipLaunchKernelGGL((sgd1d_team_fused_kernel<B, true, false>), grid, block, lds, st, a, d_its, n_iters, pool);
        else            hipLaunchKernelGGL((sgd1d_team_fused_kernel<B, false, false>), grid, block, 0, st, a, d_its, n_iters, pool);
    }
    return hipGetLastError();
}
// fused range of iterations; only for the team kernel with its widest bundles (what the auto policy picks
// on graphs large enough for launch overhead to matter).  pool: zeroed counters, pool_bytes(n_iters) of them, or null
// (fixed quota per wave, free-running).
size_t pool_bytes(uint64_t n_iters) { return (size_t)n_iters * POOL_SLOTS * POOL_STRIDE * sizeof(uint32_t); }
hipError_t launch_1d_fused(const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables, uint32_t *pool,
                           dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    switch (a.bundle) {
        case 16: return launch_1d_fused_b<16>(a, d_its, n_iters, lds_tables, pool, grid, block, lds, st);
        case 32: return launch_1d_fused_b<32>(a, d_its, n_iters, lds_tables, pool, grid, block, lds, st);
        case 64: return launch_1d_fused_b<64>(a, d_its, n_iters, lds_tables, pool, grid, block, lds, st);
        default: return hipErrorInvalidValue;
    }
}

// reference streams, fused (K1d); pool: zeroed counters, pool_bytes(n_iters) of them
hipError_t launch_1d_ref_fused(const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables, uint32_t *pool,
                               dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (lds_tables) hipLaunchKernelGGL((sgd1d_fused_kernel<true>), grid, block, lds, st, a, d_its, n_iters, pool);
    else            hipLaunchKernelGGL((sgd1d_fused_kernel<false>), grid, block, 0, st, a, d_its, n_iters, pool);
    return hipGetLastError();
}

// The first launch of a kernel function costs the host ~0.1 ms (the runtime materialises the function lazily); a
// caller that brackets its launch with events pays that inside the bracket.  Resolve the fused kernel a context will
// use when the context is set up instead — and report how many of its workgroups one CU holds at once (registers,
// waves per SIMD and the LDS table all count): the fused kernel has no grid barrier and must be launched with every
// workgroup resident (capi.hip setup_common).
template <int B>
static hipError_t prepare_1d_fused_b(bool lds_tables, int block, size_t lds, int *blocks_per_cu) {
    hipFuncAttributes attr;
    const void *fn = lds_tables ? reinterpret_cast<const void *>(&sgd1d_team_fused_kernel<B, true, true>)
                                : reinterpret_cast<const void *>(&sgd1d_team_fused_kernel<B, false, true>);
    hipError_t e = hipFuncGetAttributes(&attr, fn);
    if (e != hipSuccess) return e;
    return lds_tables ? hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, sgd1d_team_fused_kernel<B, true, true>, block, lds)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, sgd1d_team_fused_kernel<B, false, true>, block, 0);
}
hipError_t prepare_1d_fused(uint32_t bundle, bool lds_tables, int block, size_t lds, int *blocks_per_cu) {
    switch (bundle) {
        case 16: return prepare_1d_fused_b<16>(lds_tables, block, lds, blocks_per_cu);
        case 32: return prepare_1d_fused_b<32>(lds_tables, block, lds, blocks_per_cu);
        case 64: return prepare_1d_fused_b<64>(lds_tables, block, lds, blocks_per_cu);
        default: *blocks_per_cu = 0; return hipSuccess;
    }
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

// loads this translation unit's code object (HIP loads modules on first use); see gfs_warmup
hipError_t warm_module_1d() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&merge_prepare_kernel));
}

}  // namespace gfs
