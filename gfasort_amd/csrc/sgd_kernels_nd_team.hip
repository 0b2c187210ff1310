// sgd_kernels_nd_team.hip — K2b / K2c: team (bundled) kernels of path_linear_sgd_layout, D = 1..3 (one launch per iteration;
// the fused pooled launch of a range of iterations for D = 2, 3 at B = 64), and the nD launch dispatcher.
#include "sgd_kernel_common.h"

namespace gfs {

hipError_t launch_nd_ref(int dims, const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                         dim3 grid, dim3 block, size_t lds, hipStream_t st);

// ------------------------------------------------------------------------------------------
// K2b: nD team kernel — the pass/trip structure of K1b for path_linear_sgd_layout.  The two end
// flips of a term (sgd.rs:1062,1071) are drawn ONCE PER RUN, by the leader's stream right after it
// sampled the leader term: every lane of the run uses the same pair.  Each term's flips are still two
// fair independent bits; what changes is, again, only the correlation between the concurrent terms of
// a run.  With the coordinates in end x dimension planes (coord_ptr) a run on one strand then reads and
// updates 8*B CONTIGUOUS bytes per side and dimension: half the atomic requests and half the coordinate
// lines of per-lane flips (measured: 1.08 -> 0.55 requests per update for D = 2, the atomic unit being
// what binds this kernel).  Node lengths come from the following step record as in K2.  Atomics are
// issued in the trip that computes them (no deferral beyond the trip: a fused or twin trip adds once per end).
// ------------------------------------------------------------------------------------------
// One trip = (slot t of the pass, trip seg of its run, colour) — as in K1b (sgd_kernels_1d.hip): long runs for B = 64
// (sgd_device.h run_trips) and two colours for jumps shorter than the run (two_colour).
struct TripND {
    uint64_t first = 0;                                   // the path's first step (wave-uniform at B = 64)
    uint32_t qa = 0, qb = 0, qc = 0;                      // this lane's steps a, b (and c, twin trip) as ranks in the path
    uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0), na = make_uint4(0, 0, 0, 0), nb = make_uint4(0, 0, 0, 0);
    uint4 rc = make_uint4(0, 0, 0, 0), nc = make_uint4(0, 0, 0, 0);    // twin trip: the second partner's step and the step after it
    // (na, nb, nc: the steps after them, for the node lengths, sgd.rs:1051-1058 — unused where the step is its path's last, so the
    // graph's last step reads the zeroed record the index keeps behind the table, capi.hip.  Whole records: loading only their position
    // words — 8 of the 16 bytes, 10 registers less — made the kernel slower, 32.9 G updates/s on C4.)
    uint32_t cnt = 0, flips = 0, k = 1, off = 0;
    bool valid = false, two = false, fused = false;   // fused: both colours of a short-jump trip in this one (fused_trip_nd)
    bool twin = false;                                // both partners of an aligned leader in this trip (twin_trip_nd)
    int mshift = 0;
};

// lflips: the run's end flips — bit 0: step a, bit 1: partner 0's step b, bit 2: partner 1's (two partners per leader,
// sgd_device.h Leader: D >= 2 at B = 64).  tr.flips = bit 0: a, bit 1: the b of the partner this trip works on, bit 2: as drawn.
// (wave-uniform by construction at B = 64 — every input is a leader value read from ONE lane — but not always for the compiler:
// said explicitly, the trip machine's control flow is scalar branches and its arithmetic scalar instructions)
template <int B> __device__ __forceinline__ uint32_t uni(uint32_t v) { return B == 64 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)v) : v; }
template <int B> __device__ __forceinline__ bool uni(bool v) { return B == 64 ? __builtin_amdgcn_readfirstlane((int)v) != 0 : v; }

template <int B, bool FUSE>
__device__ __forceinline__ void expand_trip_nd(const KArgs &a, const Leader &L, uint32_t lflips, int t, uint32_t seg, uint32_t p, uint32_t colour,
                                               int sub, int q, TripND &tr) {
    constexpr int RUNS = 64 / B;
    const int ll = t * RUNS + q;
    const uint32_t okw = bcast<B>(L.ok, ll);
    const uint32_t ok = leader_ok(okw, p);
    const uint32_t ra0 = p ? bcast<B>(L.ra1, ll) : bcast<B>(L.ra0, ll), rb0 = p ? bcast<B>(L.rb1, ll) : bcast<B>(L.rb0, ll);
    const uint32_t fl = bcast<B>(lflips, ll);
    tr.first = bcast_first<B>(L, ll); tr.cnt = bcast<B>(L.cnt, ll);
    const uint4 *recs = a.step_rec + tr.first;                     // the path's records
    tr.flips = (fl & 1u) | (((fl >> (1u + p)) & 1u) << 1) | (fl & 4u);
    // (the number of trips must not depend on the partner: the trips of a slot go seg by seg, both partners each)
    tr.k = uni<B>((B == 64 && ((okw | (okw >> 8)) & 1u) && tr.cnt >= 2u * B) ? run_trips(a.chain, (uint32_t)B, tr.cnt) : 1u);
    tr.off = uni<B>(B == 64 ? run_offset((uint32_t)B, tr.cnt, tr.k, ra0, rb0, seg) : 0u);
    tr.mshift = (int)uni<B>((uint32_t)merged_trip_shift<B>(ok, tr.cnt, ra0, rb0, tr.off));
    const bool two = uni<B>(!(a.dbg & 0x08u) && two_colour<B>(ok, tr.cnt, ra0, rb0));
    tr.two = B == 64 ? two : (__any(two) != 0);
    tr.ra = make_uint4(0, 0, 0, 0); tr.rb = tr.ra; tr.na = tr.ra; tr.nb = tr.ra; tr.rc = tr.ra; tr.nc = tr.ra;
    tr.fused = uni<B>(FUSE && B == 64 && tr.mshift != 0 && colour == 0 && two && !(a.dbg & 0x100u));
    // both partners line-aligned long jumps whose blocks keep two trips apart: their a-runs are the same blocks, one trip serves both
    // (sgd_kernels_1d.hip expand_trip has the reasons)
    tr.twin = FUSE && B == 64 && p == 0u && a.partners == 2u && (okw & 3u) == 3u && ((okw >> 8) & 3u) == 3u && !(a.dbg & 0x04u);
    if (tr.twin) {
        const uint32_t rb1 = bcast<B>(L.rb1, ll);
        const int64_t gap = (int64_t)rb0 - (int64_t)rb1, lim = 192;
        if (gap < lim && gap > -lim) tr.twin = false;
    }
    tr.twin = uni<B>(tr.twin);
    if (tr.twin) {
        {
            const uint32_t rb1 = bcast<B>(L.rb1, ll);
            tr.qa = ra0 + tr.off + (uint32_t)sub;
            tr.qb = rb0 + tr.off + (((uint32_t)sub + ((okw >> 2) & 7u)) & 63u);
            tr.qc = rb1 + tr.off + (((uint32_t)sub + ((okw >> 10) & 7u)) & 63u);
            tr.valid = true;
            tr.ra = recs[tr.qa]; tr.rb = recs[tr.qb]; tr.rc = recs[tr.qc];
            tr.na = recs[(uint64_t)tr.qa + 1u];
            tr.nb = recs[(uint64_t)tr.qb + 1u];
            tr.nc = recs[(uint64_t)tr.qc + 1u];
            return;
        }
    }
    if (tr.fused) {
        // every lane takes its own step of the trip, its partner's record and the two records after them (node lengths)
        const int dst = sub + tr.mshift;
        tr.qa = merged_trip_base(tr.cnt, ra0, tr.off) + (uint32_t)sub;
        tr.qb = (uint32_t)((int64_t)tr.qa + tr.mshift);                // inside the path (merged_trip_shift)
        tr.valid = dst < 0 || dst > 63;                                // partner beyond the trip's 64 steps
        tr.ra = recs[tr.qa]; tr.rb = recs[tr.qb];
        tr.na = recs[(uint64_t)tr.qa + 1u];
        tr.nb = recs[(uint64_t)tr.qb + 1u];
        return;
    }
    // the generic trip: this lane's own term of the run, if it has one
    uint64_t sa = 0, sb = 0;
    tr.valid = expand_run<B>(ok, tr.first, tr.cnt, ra0, rb0, sub, colour, tr.off, sa, sb);
    if (tr.valid) {
        tr.qa = (uint32_t)(sa - tr.first); tr.qb = (uint32_t)(sb - tr.first);
        tr.ra = a.step_rec[sa]; tr.rb = a.step_rec[sb];
        tr.na = a.step_rec[sa + 1u];
        tr.nb = a.step_rec[sb + 1u];
    }
}

// The adds of one trip, D >= 2: each lane brings up to two (A, B: pointer to dimension 0 of an end, D values, flag).  In the
// dimension planes (sgd_device.h coord_ptr) the lanes of a run address consecutive doubles, so an instruction of the wave is
// 8 full lines as it stands.  (Round 2 kept [end][slot][dim] and re-dealt the adds between lanes first: 12-24 lane permutes
// per trip, ~12 % of the kernel, profiles/r03/nd_ablate.log.)
template <int D, bool HAS_B = true>
__device__ __forceinline__ void issue_adds(const uint64_t cs, const double (&vA)[D], const double (&vB)[D],
                                           double *pA, double *pB, const bool fA, const bool fB) {
    if (fA) {
#pragma unroll
        for (int d = 0; d < D; ++d) add_pos(pA + d * cs, vA[d]);
    }
    if (HAS_B && fB) {
#pragma unroll
        for (int d = 0; d < D; ++d) add_pos(pB + d * cs, vB[d]);
    }
}

// FUSED short-jump trip of the layout kernel (D = 2, 3; B = 64; the trip and all its partners inside the path) — the nD
// form of K1b's fused_trip (sgd_kernels_1d.hip).  A lane's step is the a-side of its own term in one colour and the
// b-side of its neighbour's term in the other; the two roles take the end the run's flips select (sgd.rs:1062-1077), so
// a lane keeps the coordinates of its a-end and of its b-end in registers (one set when both flips agree).  Partners
// inside the trip are read from the lane that holds them; colour 1 computes on what colour 0 produced.  Same terms and same
// arithmetic as the two generic trips; as in K1b's fused_trip a lane's OWN ends take ONE add each for both colours — the
// sum of what the lane gave as an acting lane in one colour and took as a partner in the other (its registers hold
// (c - r) + r', memory receives c + (-r + r')): one instruction per dimension when the run's two flips agree, two when
// they differ, where the adds per colour were twice that — these trips are a quarter of the layout kernel's updates and
// had 0.75 atomic requests per update against a twin trip's 0.375.  Partners beyond the trip are added per colour.
// Returns false when the quota filled between the colours.
template <int D, bool ATOMIC_LOADS, bool TRACE>
__device__ __forceinline__ bool fused_trip_nd(const KArgs &a, const TripND &cur, const int lane, const uint32_t tid,
                                              const uint64_t wave_quota, uint64_t &wave_done, uint32_t &done, uint32_t &att, uint32_t &ntr) {
    const uint64_t cs = coord_step(a);
    const int s = cur.mshift, z = s < 0 ? -s : s;
    const int dst = lane + s, src = lane - s;
    const bool out = cur.valid;
    const int dstc = out ? lane : dst, srcc = (src < 0 || src > 63) ? lane : src;
    const uint32_t grp = ((cur.off + (uint32_t)lane) / (uint32_t)z) & 1u;
    const bool fa = (cur.flips & 1u) != 0u, fb = (cur.flips & 2u) != 0u;      // wave-uniform
    const uint32_t last_q = cur.cnt - 1u;
    const uint64_t plen = a.path_len[uni<64>(rec_path(cur.ra))];      // (every lane of the trip is on the leader's path)
    // my step in both roles
    const uint64_t p_own = rec_pos_u64(cur.ra), e_own = cur.qa == last_q ? plen : rec_pos_u64(cur.na);
    const bool rev_own = (cur.ra.y >> 31) != 0;
    const double len_own = (double)(e_own - p_own);
    const double pos_a = (double)p_own + (fa ? len_own : 0.0);                // sgd.rs:1047,1062-1064
    const bool end_a = fa ? !rev_own : rev_own;
    const bool end_b_own = fb ? !rev_own : rev_own;
    // my partner's step as b-side (its record is loaded whether it sits inside the trip or not)
    const uint64_t p_p = rec_pos_u64(cur.rb), e_p = cur.qb == last_q ? plen : rec_pos_u64(cur.nb);
    const bool rev_p = (cur.rb.y >> 31) != 0;
    const double pos_b = (double)p_p + (fb ? (double)(e_p - p_p) : 0.0);      // :1048,1071-1073
    const bool end_b = fb ? !rev_p : rev_p;
    const double term_dist = fabs(pos_a - pos_b);                             // :1080
    const uint32_t node = cur.ra.x, pnode = cur.rb.x;
    const bool term_ok = term_dist != 0.0 && node != 0xFFFFFFFFu && pnode != 0xFFFFFFFFu;
    const uint64_t idx_i = (uint64_t)node * 2u + (end_a ? 1u : 0u), idx_j = (uint64_t)pnode * 2u + (end_b ? 1u : 0u);
    const bool same = idx_i == idx_j;
    const int crowd = crowd_shift<true>(a, cur.ra, cur.rb);
    double *ptr_a = coord_ptr<D>(a, node == 0xFFFFFFFFu ? 0u : node, end_a);
    double *ptr_bo = coord_ptr<D>(a, node == 0xFFFFFFFFu ? 0u : node, end_b_own);
    double *ptr_p = coord_ptr<D>(a, pnode == 0xFFFFFFFFu ? 0u : pnode, end_b);
    // coordinates: my a-end, my b-end (the same registers when the flips agree), my partner's b-end when it is outside
    double ca[D], cb[D], cp[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { ca[d] = 0.0; cb[d] = 0.0; cp[d] = 0.0; }
    if (node != 0xFFFFFFFFu) {
#pragma unroll
        for (int d = 0; d < D; ++d) ca[d] = load_pos<ATOMIC_LOADS>(ptr_a + d * cs);
        if (fa != fb) {
#pragma unroll
            for (int d = 0; d < D; ++d) cb[d] = load_pos<ATOMIC_LOADS>(ptr_bo + d * cs);
        }
    }
    if (out && pnode != 0xFFFFFFFFu) {
#pragma unroll
        for (int d = 0; d < D; ++d) cp[d] = load_pos<ATOMIC_LOADS>(ptr_p + d * cs);
    }
    double acc_a[D], acc_b[D];                                        // what my a-end and my b-end have taken so far
    bool t_a = false, t_b = false, second = true;
#pragma unroll
    for (int d = 0; d < D; ++d) { acc_a[d] = 0.0; acc_b[d] = 0.0; }
#pragma unroll
    for (uint32_t colour = 0; colour < 2u; ++colour) {
        ++att;
        bool valid = term_ok && grp == colour;
        const unsigned long long vmask = __ballot(valid);
        const uint64_t remaining = wave_quota - wave_done;
        const uint32_t nvalid = (uint32_t)__popcll(vmask);
        if (valid && nvalid > remaining) valid = (uint32_t)__popcll(vmask & ((1ull << lane) - 1ull)) < remaining;
        wave_done += nvalid < remaining ? nvalid : remaining;
        // my partner's CURRENT b-end coordinates (each shuffle a statement of its own, see fused_trip)
        double cj[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double from_a = __shfl(ca[d], dstc, 64), from_b = __shfl(cb[d], dstc, 64);
            cj[d] = out ? cp[d] : (fa != fb ? from_b : from_a);
        }
        double r_d[D];
#pragma unroll
        for (int d = 0; d < D; ++d) r_d[d] = 0.0;
        if (valid) {
            double mu = crowd_scale(fmin(a.it.eta * (1.0 / term_dist), 1.0), crowd);  // :1085-1086
            double deltas[D], mag_sq = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) { deltas[d] = ca[d] - cj[d]; mag_sq += deltas[d] * deltas[d]; }   // :1108-1113
            if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }           // :1116-1119
            const double mag = sqrt(mag_sq);                                   // :1121
            const double delta = mu * (mag - term_dist) / 2.0;                 // :1125
            const double r = delta / mag;                                      // :1142
#pragma unroll
            for (int d = 0; d < D; ++d) r_d[d] = r * deltas[d];
            ++done;                                                            // :1151
            if (TRACE) {
                if (ntr < a.trace_per_stream) {
                    TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ntr;
                    tt->i = (uint32_t)idx_i; tt->j = (uint32_t)idx_j; tt->d = term_dist;
                    ++ntr;
                }
            }
        }
        // the +r of the lane whose partner I am
        double rv[D];
#pragma unroll
        for (int d = 0; d < D; ++d) rv[d] = __shfl(r_d[d], srcc, 64);
        const int vsrc = __shfl((int)valid, srcc, 64);
        const bool recv = src >= 0 && src <= 63 && vsrc != 0;
        // a lane acts (its a-end moves by -r) or receives (its b-end moves by +r) in a colour, never both; registers and the
        // sums for the adds at the end of the trip
        if (valid && !same) {
#pragma unroll
            for (int d = 0; d < D; ++d) { ca[d] = ca[d] - r_d[d]; acc_a[d] = t_a ? acc_a[d] - r_d[d] : -r_d[d]; }   // :1143-1146
            t_a = true;
        }
        if (recv) {
            if (fa != fb) {
#pragma unroll
                for (int d = 0; d < D; ++d) { cb[d] = cb[d] + rv[d]; acc_b[d] = t_b ? acc_b[d] + rv[d] : rv[d]; }   // :1147-1148
                t_b = true;
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) { ca[d] = ca[d] + rv[d]; acc_a[d] = t_a ? acc_a[d] + rv[d] : rv[d]; }
                t_a = true;
            }
        }
        // my partner's end when it lies outside the trip
        if (!(a.dbg & 1u)) issue_adds<D, false>(cs, r_d, r_d, ptr_p, ptr_p, valid && out, false);
        if (colour == 0 && wave_done >= wave_quota) { second = false; break; }
    }
    if (!(a.dbg & 1u)) issue_adds<D>(cs, acc_a, acc_b, ptr_a, ptr_bo, t_a, t_b);
    return second;
}

// TWIN trip of the layout kernel (D = 2, 3; B = 64; two partners, both line-aligned long jumps whose blocks keep two trips apart) — the nD
// form of K1b's twin_trip (sgd_kernels_1d.hip).  A lane's step a is the a-side of two terms, (a, b) and (a, c); the end of
// a is the one the run's a-flip selects in both (sgd.rs:1062-1068), the ends of b and c follow their own flips.  One load
// of a's records and coordinates serves both terms; the second computes on what the first left in the registers; a's end
// takes ONE add, -(r + r'), b's and c's one each: 3 blocks for 128 updates where two trips take 4.  Returns false when the
// quota filled before the second term (the pass is dropped in nD, and with it the second partner's term).
template <int D, bool ATOMIC_LOADS, bool TRACE>
__device__ __forceinline__ bool twin_trip_nd(const KArgs &a, const TripND &cur, const int lane, const uint32_t tid,
                                             const uint64_t wave_quota, uint64_t &wave_done, uint32_t &done, uint32_t &att, uint32_t &ntr) {
    const uint64_t cs = coord_step(a);
    const bool fa = (cur.flips & 1u) != 0u, fb = (cur.flips & 2u) != 0u, fc = (cur.flips & 4u) != 0u;     // wave-uniform
    const uint32_t last_q = cur.cnt - 1u;
    const uint64_t plen = a.path_len[uni<64>(rec_path(cur.ra))];      // (every lane of the trip is on the leader's path)
    // step a
    const uint64_t p_a = rec_pos_u64(cur.ra), e_a = cur.qa == last_q ? plen : rec_pos_u64(cur.na);
    const bool rev_a = (cur.ra.y >> 31) != 0;
    const double pos_a = (double)p_a + (fa ? (double)(e_a - p_a) : 0.0);              // sgd.rs:1047,1062-1064
    const bool end_a = fa ? !rev_a : rev_a;
    const uint32_t node = cur.ra.x;
    const uint64_t idx_a = (uint64_t)node * 2u + (end_a ? 1u : 0u);
    double *ptr_a = coord_ptr<D>(a, node == 0xFFFFFFFFu ? 0u : node, end_a);
    double ca[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { ca[d] = 0.0; acc[d] = 0.0; }
    const bool no_loads = (a.dbg & 2u) != 0u;                                          // ablation (wrong results): no coordinate loads
    if (node != 0xFFFFFFFFu) {
#pragma unroll
        for (int d = 0; d < D; ++d) ca[d] = no_loads ? (double)(node + d) : load_pos<ATOMIC_LOADS>(ptr_a + d * cs);
    }
    // the two partners: position, end, coordinates (all loaded before any add of the trip)
    double pos_p[2], cp[2][D]; double *ptr_p[2]; uint64_t idx_p[2]; uint32_t node_p[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint4 &rp = p ? cur.rc : cur.rb; const uint4 &np = p ? cur.nc : cur.nb;
        const uint32_t qp = p ? cur.qc : cur.qb;
        const bool fp = p ? fc : fb;
        const uint64_t p_p = rec_pos_u64(rp), e_p = qp == last_q ? plen : rec_pos_u64(np);
        const bool rev_p = (rp.y >> 31) != 0;
        pos_p[p] = (double)p_p + (fp ? (double)(e_p - p_p) : 0.0);                    // :1048,1071-1073
        const bool end_p = fp ? !rev_p : rev_p;
        node_p[p] = rp.x;
        idx_p[p] = (uint64_t)rp.x * 2u + (end_p ? 1u : 0u);
        ptr_p[p] = coord_ptr<D>(a, rp.x == 0xFFFFFFFFu ? 0u : rp.x, end_p);
#pragma unroll
        for (int d = 0; d < D; ++d) cp[p][d] = 0.0;
        if (rp.x != 0xFFFFFFFFu) {
#pragma unroll
            for (int d = 0; d < D; ++d) cp[p][d] = no_loads ? (double)(rp.x + 7u * d) : load_pos<ATOMIC_LOADS>(ptr_p[p] + d * cs);
        }
    }
    bool touched = false, second = true;
    double rr[2][D]; int fadd[2] = {0, 0};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int d = 0; d < D; ++d) rr[p][d] = 0.0;
        if (!second) continue;                                                         // (wave-uniform)
        ++att;
        const double term_dist = fabs(pos_a - pos_p[p]);                               // :1080
        bool valid = term_dist != 0.0 && node != 0xFFFFFFFFu && node_p[p] != 0xFFFFFFFFu;   // :1081, :1089-1096
        const unsigned long long vmask = __ballot(valid);
        const uint64_t remaining = wave_quota - wave_done;
        const uint32_t nvalid = (uint32_t)__popcll(vmask);
        if (valid && nvalid > remaining) valid = (uint32_t)__popcll(vmask & ((1ull << lane) - 1ull)) < remaining;
        wave_done += nvalid < remaining ? nvalid : remaining;
        if (valid) {
            const double mu = crowd_scale(fmin(a.it.eta * (1.0 / term_dist), 1.0), crowd_shift<true>(a, cur.ra, p ? cur.rc : cur.rb));   // :1085-1086
            double deltas[D], mag_sq = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) { deltas[d] = ca[d] - cp[p][d]; mag_sq += deltas[d] * deltas[d]; }   // :1108-1113
            if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }                   // :1116-1119
            const double mag = sqrt(mag_sq);                                           // :1121
            const double delta = mu * (mag - term_dist) / 2.0;                         // :1125
            const double r = delta / mag;                                              // :1142
            const bool same = idx_a == idx_p[p];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                rr[p][d] = r * deltas[d];
                if (!same) { ca[d] = ca[d] - rr[p][d]; acc[d] = touched ? acc[d] - rr[p][d] : -rr[p][d]; }   // :1143-1146
            }
            if (!same) touched = true;
            fadd[p] = 1;                                                               // :1147-1148
            ++done;                                                                    // :1151
            if (TRACE) {
                if (ntr < a.trace_per_stream) {
                    TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ntr;
                    tt->i = (uint32_t)idx_a; tt->j = (uint32_t)idx_p[p]; tt->d = term_dist;
                    ++ntr;
                }
            }
        }
        if (p == 0 && wave_done >= wave_quota) second = false;
    }
    if (!(a.dbg & 1u)) {
        // the adds: a's end and b's end, then c's end
        issue_adds<D>(cs, acc, rr[0], ptr_a, ptr_p[0], touched, fadd[0] != 0);
        issue_adds<D, false>(cs, rr[1], rr[1], ptr_p[1], ptr_p[1], fadd[1] != 0, false);
    }
    return second;
}

// (3 waves per SIMD, <= 168 VGPRs: 165 at D = 2, nothing spilled; D = 3: two waves, 176, see nd_waves_for.  Round 2's kernel needed ~210 and ran
// two waves — a twin trip holds the records of three steps and of the steps after them, the next trip's too, and three ends'
// coordinates.  What brought it under 168: the trip machine's state in scalar registers (uni), steps as 32-bit ranks in their
// path, no lane permutes for the adds (dimension planes), the sampler's constants re-read per pass.  Three waves hide the
// round trip of a trip's loads and adds behind two other waves' arithmetic: without the adds the kernel runs at 64 G updates/s
// where two waves gave 51 (profiles/r03/nd_waves3.log).  Hence also the bound on the workgroup size, checked by the host.)
#ifndef GFS_ND_TEAM_WAVES
#define GFS_ND_TEAM_WAVES 3
#endif
// (D = 3 holds half as many coordinates again: three waves' worth of registers spill 4-11 of them, and under the work pool two
// waves are as fast — 34.2 against 34.3 G updates/s on C4, profiles/r03/nd_k_probe_fused.log — so D = 3 is built for two.)
constexpr int nd_waves_for(int dims) { return dims >= 3 ? 2 : GFS_ND_TEAM_WAVES; }
// The wave's state across chunks, iterations and (through KArgs.lead) launches — as K1b's TeamState (sgd_kernels_1d.hip), plus the
// run's end flips.
struct NdTeamState {
    Rng rng;
    Leader L = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t lflips = 0;                                // the pass's end flips, per leader (bit 0: a, bit 1: b, bit 2: the second partner's b)
    uint32_t left = 0, cool = 0, colour = 0, seg = 0, p = 0;
    uint32_t done = 0, att = 0, ntr = 0;
};

// One CHUNK of one wave's work in an iteration: passes and trips until `wave_quota` updates are done.  Exactly K1b's
// team_iteration: a pass outlives the chunk and the iteration it was sampled in and is dropped when the cooling phase changes;
// a quota that fills between the two terms of a twin trip, or between the two colours of a fused one, leaves the second as the next
// chunk's first trip (generic form); the rank cut-off that makes a count exact applies at the end of every chunk.  (Round 2's layout
// kernel dropped what was left of a pass at the end of a launch and knew no chunks: with them an iteration can be drawn from a work
// pool, K2c below.)
template <int D, int B, bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__device__ __forceinline__ void nd_team_iteration(const KArgs &a, const uint4 *path_tab, const double *zeta_tab, NdTeamState &ts,
                                                  const uint32_t tid, const uint64_t wave_quota, const IterConsts *itp = nullptr) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & (B - 1);
    const int q = lane / B;
    const uint64_t cs = coord_step(a);
    const uint64_t max_passes = (uint64_t)a.attempt_factor * (wave_quota / (64u * B) + 1u) + 16u;
    uint64_t wave_done = 0, passes = 0;
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
            ts.lflips = ts.rng.flip() | (ts.rng.flip() << 1);             // the run's end flips: bit 0 = a, bit 1 = b
            if (a.partners == 2u) ts.lflips |= ts.rng.flip() << 2;        // bit 2 = the second partner's b
            ts.left = B; ts.cool = (uint32_t)a.it.cooling; ts.colour = 0; ts.seg = 0; ts.p = 0;
        }
        const Leader &L = ts.L;
        const uint32_t lflips = ts.lflips;
        int t = B - (int)ts.left;
        uint32_t colour = ts.colour, seg = ts.seg, p = ts.p;
        TripND cur;
        expand_trip_nd<B, (D >= 2)>(a, L, lflips, t, seg, p, colour, sub, q, cur);
        for (;;) {
            t = (int)uni<B>((uint32_t)t); seg = uni<B>(seg); colour = uni<B>(colour); p = uni<B>(p);
            const bool c_two = uni<B>(cur.two), c_fused = uni<B>(cur.fused), c_twin = uni<B>(cur.twin);
            const uint32_t c_k = uni<B>(cur.k);
            // the trip after this one (second colour, the leader's second partner, next trip of the run, next slot — as in
            // K1b, sgd_kernels_1d.hip): request its records now
            int t_n = t; uint32_t colour_n = 0u, seg_n = seg, p_n = p;
            if (colour == 0 && c_two && !c_fused) colour_n = 1u;
            else if (p == 0u && a.partners == 2u && !c_twin) p_n = 1u;
            else if (seg + 1u < c_k) { seg_n = seg + 1u; p_n = 0u; }
            else { t_n = t + 1; seg_n = 0u; p_n = 0u; }
            const bool have_n = t_n < B;
            TripND nxt;
            if (have_n) expand_trip_nd<B, (D >= 2)>(a, L, lflips, t_n, seg_n, p_n, colour_n, sub, q, nxt);
            if (D >= 2 && B == 64 && c_twin) {
                if (!twin_trip_nd<D, ATOMIC_LOADS, TRACE>(a, cur, lane, tid, wave_quota, wave_done, ts.done, ts.att, ts.ntr)) {
                    ts.colour = 0u; ts.seg = seg; ts.p = 1u;           // quota filled between the partners: the second one is
                    break;                                             // the next chunk's first trip (generic form)
                }
                ts.colour = 0u; ts.seg = seg_n; ts.p = 0u;
                if (t_n != t) --ts.left;
                if (wave_done >= wave_quota || !have_n) break;
                cur = nxt; t = t_n; colour = colour_n; seg = seg_n; p = p_n;
                continue;
            }
            if (D >= 2 && B == 64 && c_fused) {
                if (!fused_trip_nd<D, ATOMIC_LOADS, TRACE>(a, cur, lane, tid, wave_quota, wave_done, ts.done, ts.att, ts.ntr)) {
                    ts.colour = 1u; ts.seg = seg; ts.p = p;            // quota filled between the colours: the second one is
                    break;                                             // the next chunk's first trip (generic form)
                }
                ts.colour = 0u; ts.seg = seg_n; ts.p = p_n;
                if (t_n != t) --ts.left;
                if (wave_done >= wave_quota || !have_n) break;
                cur = nxt; t = t_n; colour = colour_n; seg = seg_n; p = p_n;
                continue;
            }
            // consume the current trip (generic form)
            ts.colour = colour_n; ts.seg = seg_n; ts.p = p_n;
            if (t_n != t) --ts.left;
            bool valid = cur.valid;
            const uint4 ra = cur.ra, rb = cur.rb, na = cur.na, nb = cur.nb;
            const uint32_t qa = cur.qa, qb = cur.qb;
            const uint32_t cnt = cur.cnt, flips = cur.flips;
            const int mshift = cur.mshift;
            ++ts.att;
            double term_dist = 0.0;
            uint64_t idx_i = 0, idx_j = 0;
            bool oa = false, ob = false;
            if (valid) {
                const uint32_t last_q = cnt - 1u;
                const uint64_t plen = a.path_len[rec_path(ra)];
                const uint64_t pa = rec_pos_u64(ra), pb = rec_pos_u64(rb);
                const uint64_t ea = qa == last_q ? plen : rec_pos_u64(na);
                const uint64_t eb = qb == last_q ? plen : rec_pos_u64(nb);
                double pos_a = (double)pa, pos_b = (double)pb;                         // sgd.rs:1047-1048
                const double len_i = (double)(ea - pa), len_j = (double)(eb - pb);     // :1051-1058
                const bool rev_i = (ra.y >> 31) != 0, rev_j = (rb.y >> 31) != 0;
                oa = (flips & 1u) != 0u;                                               // :1062
                if (oa) { pos_a += len_i; oa = !rev_i; } else { oa = rev_i; }
                ob = (flips & 2u) != 0u;                                               // :1071
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
            double upd_r[D]; double *upd_ci = a.x, *upd_cj = a.x; bool upd_i = false, upd_j = false;
#pragma unroll
            for (int d = 0; d < D; ++d) upd_r[d] = 0.0;
            if (valid) {
                double mu = crowd_scale(fmin(a.it.eta * (1.0 / term_dist), 1.0), crowd_shift<true>(a, ra, rb));                   // :1085-1086
                double *ci = coord_ptr<D>(a, ra.x, oa), *cj = coord_ptr<D>(a, rb.x, ob);
                double deltas[D];
                double mag_sq = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) {                                          // :1108-1113
                    deltas[d] = load_pos<ATOMIC_LOADS>(ci + d * cs) - load_pos<ATOMIC_LOADS>(cj + d * cs);
                    mag_sq += deltas[d] * deltas[d];
                }
                if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }               // :1116-1119
                double mag = sqrt(mag_sq);                                             // :1121
                double delta = mu * (mag - term_dist) / 2.0;                           // :1125
                double r = delta / mag;                                                // :1142
                const bool same = idx_i == idx_j;
                if (D < 2) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {                                      // :1143-1149
                        double r_d = r * deltas[d];
                        if (!same) add_pos(ci + d * cs, -r_d);
                        add_pos(cj + d * cs, r_d);
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < D; ++d) upd_r[d] = r * deltas[d];
                    upd_ci = ci; upd_cj = cj; upd_i = !same; upd_j = true;
                }
                ++ts.done;                                                                // :1151
                if (TRACE) {
                    if (ts.ntr < a.trace_per_stream) {
                        TraceTerm *tt = reinterpret_cast<TraceTerm *>(a.trace) + (size_t)tid * a.trace_per_stream + ts.ntr;
                        tt->i = (uint32_t)idx_i; tt->j = (uint32_t)idx_j; tt->d = term_dist;
                        ++ts.ntr;
                    }
                }
            }
            if (D >= 2) {
                // the two adds of this lane: A = -r to end i, B = +r to end j (:1143-1149)
                double vA[D], vB[D];
#pragma unroll
                for (int k = 0; k < D; ++k) { vA[k] = -upd_r[k]; vB[k] = upd_r[k]; }
                unsigned long long pA = (unsigned long long)upd_ci, pB = (unsigned long long)upd_cj;
                int fA = (int)upd_i, fB = (int)upd_j;
                if (B == 64 && mshift != 0) {
                    // merged short-jump trip (sgd_kernel_common.h merged_trip_shift): the resting lane that sits on the
                    // partner step of an acting lane issues that lane's +r, so that one instruction carries the trip
                    const int z = mshift < 0 ? -mshift : mshift;
                    const int src = lane - mshift;
                    const int srcc = src < 0 ? 0 : (src > 63 ? 63 : src);
                    const unsigned long long pjs = __shfl(pB, srcc, 64);
                    const int fjs = __shfl(fB, srcc, 64);
                    double rs[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) rs[k] = __shfl(upd_r[k], srcc, 64);
                    const bool resting = (((cur.off + (uint32_t)lane) / (uint32_t)z) & 1u) != colour;
                    if (resting && src >= 0 && src < 64 && fjs) {
                        pA = pjs; fA = 1;
#pragma unroll
                        for (int k = 0; k < D; ++k) vA[k] = rs[k];
                    }
                    const int dst = lane + mshift;
                    fB = fB && (dst < 0 || dst > 63);                                  // partner beyond the run: add it myself
                }
                issue_adds<D>(cs, vA, vB, reinterpret_cast<double *>(pA), reinterpret_cast<double *>(pB), fA != 0, fB != 0);
            }
            if (wave_done >= wave_quota) break;                        // what is left of the pass serves the next chunk
            if (!have_n) break;
            cur = nxt; t = t_n; colour = colour_n; seg = seg_n; p = p_n;
        }
    }
}

// lead word (KArgs.lead[5]): as K1b's (sgd_kernels_1d.hip load_pass), with the pass's end flips in bits 5..7
__device__ __forceinline__ void load_pass_nd(const KArgs &a, uint32_t tid, NdTeamState &ts) {
    if (!a.lead) return;
    const uint64_t T = a.n_streams;
    ts.L.first_lo = a.lead[tid]; ts.L.first_hi = a.lead[T + tid]; ts.L.cnt = a.lead[2 * T + tid];
    ts.L.ra0 = a.lead[3 * T + tid]; ts.L.rb0 = a.lead[4 * T + tid];
    const uint32_t w = a.lead[5 * T + tid];
    ts.L.ra1 = a.lead[6 * T + tid]; ts.L.rb1 = a.lead[7 * T + tid];
    ts.L.ok = (w & 0x1Fu) | ((w >> 27) << 8);
    ts.lflips = (w >> 5) & 7u;
    const uint32_t ws = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);      // the place in the pass: the same for the whole wave
    ts.left = (ws >> 8) & 0xFFu; ts.cool = (ws >> 16) & 1u; ts.colour = (ws >> 17) & 1u; ts.seg = (ws >> 18) & 0xFFu; ts.p = (ws >> 26) & 1u;
}
__device__ __forceinline__ void store_pass_nd(const KArgs &a, uint32_t tid, const NdTeamState &ts) {
    if (!a.lead) return;
    const uint64_t T = a.n_streams;
    a.lead[tid] = ts.L.first_lo; a.lead[T + tid] = ts.L.first_hi; a.lead[2 * T + tid] = ts.L.cnt;
    a.lead[3 * T + tid] = ts.L.ra0; a.lead[4 * T + tid] = ts.L.rb0;
    a.lead[5 * T + tid] = (ts.L.ok & 0x1Fu) | ((ts.lflips & 7u) << 5) | (ts.left << 8) | (ts.cool << 16) | (ts.colour << 17) | (ts.seg << 18) |
                          (ts.p << 26) | (((ts.L.ok >> 8) & 0x1Fu) << 27);
    a.lead[6 * T + tid] = ts.L.ra1; a.lead[7 * T + tid] = ts.L.rb1;
}

// K2b: one launch = one iteration, a fixed quota per wave worked through in chunks (like a pool of its own: one wave is bit for
// bit the oracle's mirror here and in K2c).
template <int D, int B, bool LDS_TABLES, bool ATOMIC_LOADS, bool TRACE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(nd_waves_for(D), nd_waves_for(D)))) sgdnd_team_kernel(const KArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint4 *path_tab; const double *zeta_tab;
    stage_tables<LDS_TABLES>(a, smem, path_tab, zeta_tab);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= a.n_streams) return;
    const uint64_t T = a.n_streams;
    NdTeamState ts;
    ts.rng.s0 = a.rng[tid]; ts.rng.s1 = a.rng[T + tid]; ts.rng.s2 = a.rng[2 * T + tid]; ts.rng.s3 = a.rng[3 * T + tid];
    ts.ntr = TRACE ? a.trace_cnt[tid] : 0;
    load_pass_nd(a, tid, ts);
    // (readfirstlane: the wave's quota, and with it every loop variable of the trip machine, is then wave-uniform for the
    // compiler too — scalar registers and scalar arithmetic instead of 64 copies, as in K1c)
    const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid & ~63u));
    uint64_t wq = (uint64_t)a.quota_base * 64u;
    if (wave_first < a.quota_rem) wq += (a.quota_rem - wave_first) < 64u ? (a.quota_rem - wave_first) : 64u;
    for (uint64_t done = 0; done < wq; done += a.chunk)
        nd_team_iteration<D, B, LDS_TABLES, ATOMIC_LOADS, TRACE>(a, path_tab, zeta_tab, ts, tid, wq - done < a.chunk ? wq - done : a.chunk);
    // (the stream's addresses are computed again here rather than kept in registers since the loads at the top: built for three
    // waves per SIMD the kernel would otherwise spill exactly those registers, and a kernel with scratch pays for its set-up)
    uint32_t tid_out = tid;
    asm volatile("" : "+v"(tid_out));
    a.rng[tid_out] = ts.rng.s0; a.rng[T + tid_out] = ts.rng.s1; a.rng[2 * T + tid_out] = ts.rng.s2; a.rng[3 * T + tid_out] = ts.rng.s3;
    if (TRACE) a.trace_cnt[tid] = ts.ntr;
    store_pass_nd(a, tid_out, ts);
    flush_counters(a, ts.done, ts.att);
}

// K2c: the same, FUSED over a range of iterations with WORK POOLS — K1c (sgd_kernels_1d.hip, which has the reasons) for layouts.
// Fixed quotas leave a layout launch's waves finishing up to a fifth of the launch apart (a wave's trips cost by what its few
// leaders happen to be; profiles/r03/nd_k_probe.log: the longer the runs, the fewer leaders per wave and the slower the launch);
// drawn from a pool, an iteration ends for all waves within a chunk of each other, and the next one starts without a launch.
template <int D, int B, bool LDS_TABLES, bool POOL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(nd_waves_for(D), nd_waves_for(D))))
sgdnd_team_fused_kernel(const KArgs a0, const IterConsts *its, const uint32_t n_iters, uint32_t *pool) {
    constexpr bool ATOMIC_LOADS = true;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint4 *path_tab; const double *zeta_tab;
    stage_tables<LDS_TABLES>(a0, smem, path_tab, zeta_tab);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= a0.n_streams) return;
    const uint64_t T = a0.n_streams;
    KArgs a = a0;
    NdTeamState ts;
    ts.rng.s0 = a.rng[tid]; ts.rng.s1 = a.rng[T + tid]; ts.rng.s2 = a.rng[2 * T + tid]; ts.rng.s3 = a.rng[3 * T + tid];
    const int lane = threadIdx.x & 63;
    load_pass_nd(a, tid, ts);
    if (POOL) {
        // ONE counter per iteration for layouts.  K1c spreads an iteration over up to 16 counters so that the claims do not queue on
        // one address; each is a fixed share of the iteration, so the waves of a fast counter run ahead of the others' — without
        // bound over a schedule.  The sort does not notice; a layout does: on DRB1 x120 the median |distance between a node's two
        // ends - its length| was 1.6 bp with 16 counters, 1.15 with one, 1.09 with a launch per iteration and 2.4 with free-running
        // waves (reference streams 0.92; profiles/r03/tiled_layout_e2e_probe.log, nd_pool_slots_probe.log).  A layout chunk is
        // 16 heavy trips, so one counter takes ~2e7 claims/s at most: 49.5 against 50.3 G updates/s on C4.
        constexpr uint32_t slot = 0u;
        const uint32_t cap = (uint32_t)((uint64_t)a0.quota_base * a0.n_streams + a0.quota_rem);   // the iteration's updates, < 2^31 (host-checked)
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
            nd_team_iteration<D, B, LDS_TABLES, ATOMIC_LOADS, false>(a, path_tab, zeta_tab, ts, tid, cap - old < a0.chunk ? cap - old : a0.chunk, its + k);
        }
    } else {
        // fixed quota per wave and iteration, free-running (GFS_F_DBG_FREE_RUNNING)
        const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid & ~63u));
        uint64_t wq = (uint64_t)a.quota_base * 64u;
        if (wave_first < a.quota_rem) wq += (a.quota_rem - wave_first) < 64u ? (a.quota_rem - wave_first) : 64u;
        for (uint32_t k = 0; k < n_iters; ++k) {
            a.it = its[k];
            for (uint64_t done = 0; done < wq; done += a.chunk)
                nd_team_iteration<D, B, LDS_TABLES, ATOMIC_LOADS, false>(a, path_tab, zeta_tab, ts, tid, wq - done < a.chunk ? wq - done : a.chunk, its + k);
        }
    }
    uint32_t tid_out = tid;
    asm volatile("" : "+v"(tid_out));
    a.rng[tid_out] = ts.rng.s0; a.rng[T + tid_out] = ts.rng.s1; a.rng[2 * T + tid_out] = ts.rng.s2; a.rng[3 * T + tid_out] = ts.rng.s3;
    store_pass_nd(a, tid_out, ts);
    flush_counters(a, ts.done, ts.att);
}

template <int D, int B>
static hipError_t launch_ndb(const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                             dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    // team kernel variants: LDS tables on/off; agent-scope loads; trace on/off
    (void)atomic_loads;                               // team layout kernels always use agent-scope loads
    if (lds_tables) {
        if (trace) hipLaunchKernelGGL((sgdnd_team_kernel<D, B, true, true, true>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((sgdnd_team_kernel<D, B, true, true, false>), grid, block, lds, st, a);
    } else {
        if (trace) hipLaunchKernelGGL((sgdnd_team_kernel<D, B, false, true, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((sgdnd_team_kernel<D, B, false, true, false>), grid, block, 0, st, a);
    }
    return hipGetLastError();
}
hipError_t launch_nd(int dims, const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                     dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (dims >= 1 && dims <= 3 && a.bundle >= 8) {
#define GFS_NDB_CASE(D, B) if (dims == D && a.bundle == B) return launch_ndb<D, B>(a, lds_tables, atomic_loads, trace, grid, block, lds, st);
        GFS_NDB_CASE(1, 8) GFS_NDB_CASE(1, 16) GFS_NDB_CASE(1, 32) GFS_NDB_CASE(1, 64)
        GFS_NDB_CASE(2, 8) GFS_NDB_CASE(2, 16) GFS_NDB_CASE(2, 32) GFS_NDB_CASE(2, 64)
        GFS_NDB_CASE(3, 8) GFS_NDB_CASE(3, 16) GFS_NDB_CASE(3, 32) GFS_NDB_CASE(3, 64)
#undef GFS_NDB_CASE
    }
    return launch_nd_ref(dims, a, lds_tables, atomic_loads, trace, grid, block, lds, st);
}

// K2c launchers: layouts of 2 and 3 dimensions at B = 64 (what the auto policy picks on graphs large enough for it to matter).
// pool: zeroed counters, pool_bytes(n_iters) of them, or null (fixed quota per wave, free-running).
template <int D>
static hipError_t launch_nd_team_fused_d(const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables,
                                         uint32_t *pool, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (pool) {
        if (lds_tables) hipLaunchKernelGGL((sgdnd_team_fused_kernel<D, 64, true, true>), grid, block, lds, st, a, d_its, n_iters, pool);
        else            hipLaunchKernelGGL((sgdnd_team_fused_kernel<D, 64, false, true>), grid, block, 0, st, a, d_its, n_iters, pool);
    } else {
        if (lds_tables) hipLaunchKernelGGL((sgdnd_team_fused_kernel<D, 64, true, false>), grid, block, lds, st, a, d_its, n_iters, pool);
        else            hipLaunchKernelGGL((sgdnd_team_fused_kernel<D, 64, false, false>), grid, block, 0, st, a, d_its, n_iters, pool);
    }
    return hipGetLastError();
}
hipError_t launch_nd_team_fused(int dims, const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables, uint32_t *pool,
                                dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    if (a.bundle != 64u) return hipErrorInvalidValue;
    if (dims == 2) return launch_nd_team_fused_d<2>(a, d_its, n_iters, lds_tables, pool, grid, block, lds, st);
    if (dims == 3) return launch_nd_team_fused_d<3>(a, d_its, n_iters, lds_tables, pool, grid, block, lds, st);
    return hipErrorInvalidValue;
}
// workgroups of the fused kernel one CU holds at once (0: no fused kernel for this shape)
template <int D>
static hipError_t prepare_nd_team_fused_d(bool lds_tables, int block, size_t lds, int *blocks_per_cu) {
    return lds_tables ? hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, sgdnd_team_fused_kernel<D, 64, true, true>, block, lds)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, sgdnd_team_fused_kernel<D, 64, false, true>, block, 0);
}
hipError_t prepare_nd_team_fused(int dims, uint32_t bundle, bool lds_tables, int block, size_t lds, int *blocks_per_cu) {
    *blocks_per_cu = 0;
    if (bundle != 64u) return hipSuccess;
    if (dims == 2) return prepare_nd_team_fused_d<2>(lds_tables, block, lds, blocks_per_cu);
    if (dims == 3) return prepare_nd_team_fused_d<3>(lds_tables, block, lds, blocks_per_cu);
    return hipSuccess;
}

// waves per SIMD the layout team kernels are built for (the host sizes the stream count by it)
int nd_team_waves(int dims) { return nd_waves_for(dims); }

// loads this translation unit's code object (HIP loads modules on first use); see gfs_warmup
hipError_t warm_module_nd_team() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&sgdnd_team_kernel<2, 64, true, true, false>));
}

}  // namespace gfs
