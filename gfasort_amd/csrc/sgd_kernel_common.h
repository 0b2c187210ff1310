// sgd_kernel_common.h — shared pieces of the SGD batch kernels (gfx950 / CDNA4, wave64).
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
#pragma once
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

constexpr uint32_t COUNTER_SLOTS = 1024;     // counters: [COUNTER_SLOTS][8] u64, [s][0] = updates, [s][1] = attempts

__device__ __forceinline__ void flush_counters(const KArgs &a, uint32_t done, uint32_t att) {
    // wave64 butterfly, one atomic per wave
    unsigned long long d = done, t = att;
    for (int off = 32; off > 0; off >>= 1) {
        d += __shfl_xor(d, off, 64);
        t += __shfl_xor(t, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        // one 64-byte line per slot, waves spread over COUNTER_SLOTS lines: thousands of atomics on ONE
        // address serialise at the memory side and showed up as tens of microseconds at the end of every launch
        const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        unsigned long long *slot = a.counters + (size_t)(wave & (COUNTER_SLOTS - 1)) * 8;
        atomicAdd(slot, d);
        atomicAdd(slot + 1, t);
    }
}

// The launch constants of the kernel, read afresh from the kernel-argument segment (every kernel here takes its KArgs first,
// by value: offset 0).  The team kernels' sampler calls this once per pass (~1000 trips): scalar loads, and the ~40 scalar
// registers of constants only the sampler needs are then free while the trips run — held through them they were spilled to
// vector lanes and read back inside the trip machine (74 / 61 spilled in K2b / K1c).  The empty asm hides the pointer's
// origin, or the loads would be hoisted and kept live like the by-value copy.
__device__ __forceinline__ void reload_kargs(KArgs &as) {
    auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    typedef const __attribute__((address_space(4))) uint32_t kword;
    kword *kw = (kword *)kp;
    uint32_t *dw = reinterpret_cast<uint32_t *>(&as);
    static_assert(sizeof(KArgs) % 4 == 0, "KArgs is copied by words");
#pragma unroll
    for (unsigned i = 0; i < sizeof(KArgs) / 4; ++i) dw[i] = kw[i];
}

template <int B>
__device__ __forceinline__ uint32_t bcast(uint32_t v, int leader_lane) {
    if (B == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, leader_lane);     // wave-uniform leader
    return (uint32_t)__shfl((int)v, leader_lane, 64);
}


template <int B>
__device__ __forceinline__ uint64_t bcast_first(const Leader &L, int leader_lane) {
    return ((uint64_t)bcast<B>(L.first_hi, leader_lane) << 32) | bcast<B>(L.first_lo, leader_lane);
}

// Short-jump trips of a 64-lane run (|jump| < 64).  Only the lanes of every other group of |jump| lanes act
// (node-disjoint rule), and the partner of an acting lane is the step of a resting lane |jump| places on: both
// sides of the trip touch the SAME lines.  Issued as two instructions they are two requests per line for half a
// wave of updates, one straight after the other on the same lines — measured, these trips were 9 % of the trips
// and 16 % of the time.  When the whole trip lies inside the path (no wrap, no mirrored jump) the +r of a term is
// therefore handed to the resting lane that sits on its node, and ONE instruction carries every add of the trip;
// only partners beyond the run's ends are added by a second, nearly empty one.  Returns the signed jump, or 0.
template <int B>
__device__ __forceinline__ int merged_trip_shift(uint32_t ok, uint32_t cnt, uint32_t ra0, uint32_t rb0, uint32_t off) {
    if (B != 64 || (ok & 3u) != 1u || cnt < 128u) return 0;
    const int64_t s = (int64_t)rb0 - (int64_t)ra0;
    if (s == 0 || s >= 64 || s <= -64) return 0;
    uint64_t base = (uint64_t)ra0 + (uint64_t)off;                                     // first step of this trip
    if (base >= cnt) base -= cnt;
    if (base + 64u > cnt) return 0;                                                    // the trip would wrap
    if (s > 0 ? base + 63u + (uint64_t)s > (uint64_t)cnt - 1u : (int64_t)base + s < 0) return 0;
    return (int)s;
}
// first step of the trip that merged_trip_shift accepted (rank in the path)
__device__ __forceinline__ uint32_t merged_trip_base(uint32_t cnt, uint32_t ra0, uint32_t off) {
    uint64_t base = (uint64_t)ra0 + (uint64_t)off;
    if (base >= cnt) base -= cnt;
    return (uint32_t)base;
}

// ------------------------------------------------------------------------------------------
// WORK POOLS of a fused launch (K1c sgd1d_team_fused_kernel explains them; K1d / K2d below use the same counters): an
// iteration's updates are claimed in chunks from one of up to POOL_SLOTS counters, one 64-B line each.
constexpr uint32_t POOL_SLOTS = 16, POOL_STRIDE = 16;              // counters per iteration; u32 per 64-B line
// Counters in use: one per 16 waves, at most POOL_SLOTS.  Several counters exist so that 4 000 waves do not queue on one
// address; a counter must still be SHARED by many waves — a wave with a counter of its own has a fixed quota again and drifts
// away from the others in the schedule (a 40-lane last wave beside 15 full ones, each on its own counter, ran 60 % behind and
// cost DRB1 a fifth of its final stress: 0.39 against 0.33, round 3).
__host__ __device__ __forceinline__ uint32_t pool_slots(uint32_t n_waves) {
    const uint32_t s = n_waves / 16u;
    return s < 1u ? 1u : (s > POOL_SLOTS ? POOL_SLOTS : s);
}

// K1d / K2d: REFERENCE STREAMS, a range of iterations in ONE persistent launch.  The reference's workers never stop at an
// iteration boundary — the checker thread switches eta / theta / cooling under them (sgd.rs:366-403) — and they share ONE
// count of term updates per iteration (sgd.rs:579-583).  One launch per iteration costs a small graph more than its
// updates do (DRB1: 35 059 updates in 0.1 ms, most of it launch ramp and tail).  Here every wave walks the schedule
// its[0..n) and claims an iteration's updates from the pool in chunks of REF_CHUNK_PER_LANE per live lane (one returning
// atomic per wave and chunk, the next claim travelling while the chunk is worked on); a chunk is dealt to the lanes —
// each an ordinary reference stream — in equal shares.  Every iteration applies exactly min_term_updates updates under
// its own constants; no wave is more than two chunks from the others.  ONE stream claims every chunk itself, in order,
// and is bit for bit the per-iteration kernel and the oracle's single stream (tested).
constexpr uint32_t REF_CHUNK_PER_LANE = 16;

// run(share, max_attempts): the stream's loop for `share` successful updates (ref_run_1d / ref_run_nd)
template <class Run>
__device__ __forceinline__ void ref_pooled_walk(KArgs &a, const IterConsts *its, const uint32_t n_iters, uint32_t *pool,
                                                const uint32_t tid, Run &&run) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_first = tid & ~63u;                                            // < n_streams (caller)
    const uint32_t nl = a.n_streams - wave_first < 64u ? a.n_streams - wave_first : 64u;   // live lanes of this wave
    const uint32_t chunk = nl * a.ref_chunk;
    const uint32_t wave = tid >> 6, n_waves = (a.n_streams + 63u) >> 6;
    const uint32_t slots = pool_slots(n_waves), slot = wave % slots;
    const uint64_t total = (uint64_t)a.quota_base * a.n_streams + a.quota_rem;
    const uint32_t cap = (uint32_t)(total / slots + (slot < total % slots ? 1u : 0u));     // < 2^31 (host-checked)
    uint32_t k = 0, claim = 0;
    a.it = its[0];
    if (lane == 0) claim = __hip_atomic_fetch_add(pool + slot * POOL_STRIDE, chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (k < n_iters) {
        const uint32_t old = (uint32_t)__builtin_amdgcn_readfirstlane((int)claim);
        if (old >= cap) {                                                              // this iteration's pool is exhausted
            if (++k == n_iters) break;
            a.it = its[k];                                                             // wave-uniform: scalar loads
            if (lane == 0) claim = __hip_atomic_fetch_add(pool + ((size_t)k * POOL_SLOTS + slot) * POOL_STRIDE, chunk,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (lane == 0) claim = __hip_atomic_fetch_add(pool + ((size_t)k * POOL_SLOTS + slot) * POOL_STRIDE, chunk,
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t m = cap - old < chunk ? cap - old : chunk;
        const uint32_t share = lane < nl ? m / nl + (lane < m % nl ? 1u : 0u) : 0u;
        if (share) run(share, (uint64_t)a.attempt_factor * share + 64u);
    }
}

}  // namespace gfs
