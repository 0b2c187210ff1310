// Microbenchmark: rates of scattered 8-byte memory operations on MI355X, to price the
// position-update step of the SGD kernel.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint64_t xs(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

template <int MODE>
__global__ void k(double *x, const uint4 *rec, uint64_t n, uint64_t nrec, int iters, double *sink) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t s = 0x9E3779B97F4A7C15ull * (tid + 1);
    double acc = 0.0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        uint64_t r = xs(s);
        uint64_t idx = (r >> 11) % n;
        if (MODE == 6) {            // lane pairs share one 16-B slot
            uint64_t base = __shfl(idx, lane & ~1, 64);
            idx = (base & ~1ull) | (lane & 1);
        } else if (MODE == 7) {     // 8 lanes share one 64-B line
            uint64_t base = __shfl(idx, lane & ~7, 64);
            idx = (base & ~7ull) | (lane & 7);
        } else if (MODE == 14 || MODE == 15 || MODE == 16) {   // 64 lanes = a random permutation of 64 slots in 8 lines
            uint64_t base = __shfl(idx, 0, 64);
            uint32_t perm = (uint32_t)((lane * 37u + 11u) & 63u);      // bijection on 0..63 (37 odd)
            idx = ((base & ~63ull) | perm) % n;
        } else if (MODE == 17) {    // 64 lanes = 64 CONSECUTIVE slots (8 lines), in order
            uint64_t base = __shfl(idx, 0, 64);
            idx = ((base & ~63ull) | (uint64_t)lane) % n;
        } else if (MODE == 12) {    // 2 adjacent lanes in the same 64-B line but random slots
            uint64_t base = __shfl(idx, lane & ~1, 64);
            idx = (base & ~7ull) | ((idx + (lane & 1) * 3) & 7);
        }
        double v = 1e-9 * (double)(r & 255);
        if (MODE == 0 || MODE == 6 || MODE == 7 || MODE == 12 || MODE == 14 || MODE == 17)
            (void)__hip_atomic_fetch_add(x + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1)
            (void)__hip_atomic_fetch_add(x + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 2)
            (void)__hip_atomic_fetch_add(x + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else if (MODE == 3)
            (void)__hip_atomic_fetch_add(x + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (MODE == 4)
            x[idx] = v;
        else if (MODE == 5)
            __hip_atomic_store(x + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 8 || MODE == 15)
            acc += x[idx];
        else if (MODE == 16) {
            uint4 q = rec[(((r >> 11) % nrec) & ~63ull) | (uint64_t)lane];
            acc += (double)(q.x ^ q.y ^ q.z ^ q.w);
        }
        else if (MODE == 9)
            acc += __hip_atomic_load(x + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 10) {
            uint4 q = rec[(r >> 11) % nrec];
            acc += (double)(q.x ^ q.y ^ q.z ^ q.w);
        } else if (MODE == 11)
            acc += __hip_atomic_fetch_add(x + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 13) {      // load + plain store (the reference's racy read-modify-write)
            double o = __hip_atomic_load(x + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(x + idx, o + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 123.456) sink[0] = acc;
}

template <int MODE>
double run(const char *name, double *x, uint4 *rec, uint64_t n, uint64_t nrec, double *sink, int blocks, int iters) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, x, rec, n, nrec, 4, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, x, rec, n, nrec, iters, sink);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)blocks * 256 * iters;
    printf("%-44s n=%-10llu %8.3f ms  %8.2f G ops/s\n", name, (unsigned long long)n, ms, ops / ms / 1e6);
    fflush(stdout);
    return ops / ms / 1e6;
}

int main() {
    const uint64_t nrec = 10000000;
    uint4 *rec; CHK(hipMalloc(&rec, nrec * 16)); CHK(hipMemset(rec, 1, nrec * 16));
    double *sink; CHK(hipMalloc(&sink, 8));
    for (uint64_t n : {1000000ull, 10000000ull, 100000ull}) {
        double *x; CHK(hipMalloc(&x, n * 8)); CHK(hipMemset(x, 0, n * 8));
        const int blocks = 2048, iters = 64;
        run<0>("atomic add f64 agent, scattered", x, rec, n, nrec, sink, blocks, iters);
        run<1>("atomic add f64 workgroup scope", x, rec, n, nrec, sink, blocks, iters);
        run<2>("atomic add f64 wavefront scope", x, rec, n, nrec, sink, blocks, iters);
        run<3>("atomic add f64 system scope", x, rec, n, nrec, sink, blocks, iters);
        run<11>("atomic add f64 agent WITH return", x, rec, n, nrec, sink, blocks, iters);
        run<6>("atomic add, lane pairs share a 16-B slot", x, rec, n, nrec, sink, blocks, iters);
        run<12>("atomic add, lane pairs share a 64-B line", x, rec, n, nrec, sink, blocks, iters);
        run<7>("atomic add, 8 lanes per 64-B line", x, rec, n, nrec, sink, blocks, iters);
        run<14>("atomic add, 64 lanes permuted over 8 lines", x, rec, n, nrec, sink, blocks, iters);
        run<17>("atomic add, 64 lanes consecutive (8 lines)", x, rec, n, nrec, sink, blocks, iters);
        run<15>("plain load, 64 lanes permuted over 8 lines", x, rec, n, nrec, sink, blocks, iters);
        run<16>("16-B record load, 64 consecutive records", x, rec, n, nrec, sink, blocks, iters);
        run<4>("plain store 8 B scattered", x, rec, n, nrec, sink, blocks, iters);
        run<5>("agent-scope (sc1) store 8 B scattered", x, rec, n, nrec, sink, blocks, iters);
        run<13>("sc1 load + sc1 store (racy RMW)", x, rec, n, nrec, sink, blocks, iters);
        run<8>("plain load 8 B scattered", x, rec, n, nrec, sink, blocks, iters);
        run<9>("agent-scope (sc1) load 8 B scattered", x, rec, n, nrec, sink, blocks, iters);
        run<10>("plain load 16 B from 160 MB records", x, rec, n, nrec, sink, blocks, iters);
        CHK(hipFree(x));
    }
    return 0;
}
