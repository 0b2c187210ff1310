// FETCH_SIZE / WRITE_SIZE calibration on the SGD kernels' own access patterns (run under
// rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE).  Each kernel moves a KNOWN number of bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ uint64_t xs(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

// A: runs of 64 consecutive 16-B records at random run starts (team kernel record loads)
__global__ void calib_rec_runs(const uint4 *rec, uint64_t nrec, int iters, double *sink) {
    uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    uint64_t s = 0x9E3779B97F4A7C15ull * (wave + 1);
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t start = (xs(s) >> 11) % (nrec - 64);
        uint4 q = rec[start + lane];
        acc += (double)(q.x ^ q.w);
    }
    if (acc == 1.2345) sink[0] = acc;
}
// B: scattered single 16-B records (reference-stream kernel record loads)
__global__ void calib_rec_scatter(const uint4 *rec, uint64_t nrec, int iters, double *sink) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t s = 0x9E3779B97F4A7C15ull * (tid + 1);
    double acc = 0;
    for (int it = 0; it < iters; ++it) { uint4 q = rec[(xs(s) >> 11) % nrec]; acc += (double)(q.x ^ q.w); }
    if (acc == 1.2345) sink[0] = acc;
}
// C: plain streaming copy-read of the whole array, 16 B per lane (the guide's calibrated case)
__global__ void calib_stream(const uint4 *rec, uint64_t nrec, double *sink) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, n = gridDim.x * (uint64_t)blockDim.x;
    double acc = 0;
    for (uint64_t k = tid; k < nrec; k += n) { uint4 q = rec[k]; acc += (double)(q.x ^ q.w); }
    if (acc == 1.2345) sink[0] = acc;
}
// D: f64 atomic adds, runs of 64 consecutive doubles at random starts (team kernel atomics)
__global__ void calib_atomic_runs(double *x, uint64_t n, int iters) {
    uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    uint64_t s = 0x9E3779B97F4A7C15ull * (wave + 1);
    for (int it = 0; it < iters; ++it) {
        uint64_t start = (xs(s) >> 11) % (n - 64);
        (void)__hip_atomic_fetch_add(x + start + lane, 1e-9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// E: scattered f64 atomic adds
__global__ void calib_atomic_scatter(double *x, uint64_t n, int iters) {
    uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t s = 0x9E3779B97F4A7C15ull * (tid + 1);
    for (int it = 0; it < iters; ++it)
        (void)__hip_atomic_fetch_add(x + (xs(s) >> 11) % n, 1e-9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main() {
    const uint64_t nrec = 10000000, n = 1000000;
    uint4 *rec; CHK(hipMalloc(&rec, nrec * 16)); CHK(hipMemset(rec, 1, nrec * 16));
    double *x; CHK(hipMalloc(&x, n * 8)); CHK(hipMemset(x, 0, n * 8));
    double *sink; CHK(hipMalloc(&sink, 8));
    const int blocks = 1024, iters = 64;   // 262144 lanes, 4096 waves
    hipLaunchKernelGGL(calib_rec_runs, dim3(blocks), dim3(256), 0, 0, rec, nrec, iters, sink);
    hipLaunchKernelGGL(calib_rec_scatter, dim3(blocks), dim3(256), 0, 0, rec, nrec, iters, sink);
    hipLaunchKernelGGL(calib_stream, dim3(blocks), dim3(256), 0, 0, rec, nrec, sink);
    hipLaunchKernelGGL(calib_atomic_runs, dim3(blocks), dim3(256), 0, 0, x, n, iters);
    hipLaunchKernelGGL(calib_atomic_scatter, dim3(blocks), dim3(256), 0, 0, x, n, iters);
    CHK(hipDeviceSynchronize());
    printf("known bytes: rec_runs %llu rec_scatter(useful) %llu stream %llu atomic_runs %llu atomic_scatter %llu\n",
           (unsigned long long)(4096ull * iters * 1024), (unsigned long long)(262144ull * iters * 16),
           (unsigned long long)(nrec * 16), (unsigned long long)(262144ull * iters * 8), (unsigned long long)(262144ull * iters * 8));
    return 0;
}
