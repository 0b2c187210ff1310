// capi.hip — the C ABI of libgfasort_hip.so (include/gfasort_hip.h): host tables, the resident
// context (device mirror of PathIndex + positions + RNG streams) and the one-shot entry points
// that stand where path_linear_sgd / path_linear_sgd_layout stand in the reference.
#include "../../include/gfasort_hip.h"
#include "sgd_kernel_common.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

namespace gfs {
hipError_t launch_1d(const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                     dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t launch_nd(int dims, const KArgs &a, bool lds_tables, bool atomic_loads, bool trace,
                     dim3 grid, dim3 block, size_t lds, hipStream_t st);
size_t pool_bytes(uint64_t n_iters);
hipError_t launch_1d_fused(const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables, uint32_t *pool,
                           dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t launch_1d_ref_fused(const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables, uint32_t *pool,
                               dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t launch_nd_ref_fused(int dims, const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables, uint32_t *pool,
                               dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t warm_module_1d();
hipError_t prepare_1d_fused(uint32_t bundle, bool lds_tables, int block, size_t lds, int *blocks_per_cu);
hipError_t warm_module_nd();
hipError_t warm_module_nd_team();
int nd_team_waves(int dims);
hipError_t launch_nd_team_fused(int dims, const KArgs &a, const IterConsts *d_its, uint32_t n_iters, bool lds_tables, uint32_t *pool,
                                dim3 grid, dim3 block, size_t lds, hipStream_t st);
hipError_t prepare_nd_team_fused(int dims, uint32_t bundle, bool lds_tables, int block, size_t lds, int *blocks_per_cu);
hipError_t warm_module_index();
hipError_t init_positions_device(const uint32_t *d_node_len, const uint32_t *d_perm, double *d_x, uint64_t n);
hipError_t reorder_positions_device(const double *d_src, double *d_dst, const uint32_t *d_perm, uint64_t N, uint32_t D,
                                    int to_device, hipStream_t st);
hipError_t first_visit_layout_device(const uint32_t *d_step_node, uint64_t n_steps, uint64_t n_nodes, const uint64_t *d_path_first,
                                     uint32_t n_paths, uint32_t *d_perm,
                                     int *bad_out);
hipError_t build_path_index_device(const uint32_t *d_step_node, const uint8_t *d_step_is_rev, const uint32_t *d_node_len,
                                   const uint32_t *d_perm, const uint64_t *d_path_first, uint32_t n_paths,
                                   uint64_t n_steps, uint64_t n_nodes, uint64_t *d_tmp, uint4 *d_rec, uint64_t *d_path_len);
hipError_t sort_order_device(const double *d_x_layout, const uint32_t *d_perm, uint64_t n, uint64_t stride_doubles,
                             void *d_tmp, uint32_t **d_order_out);
hipError_t launch_merge_prepare(const double *x, const double *x_prev, float *buf, uint64_t n, hipStream_t st);
hipError_t launch_merge_apply(double *x, double *x_prev, const float *buf, uint64_t n, double scale_all, hipStream_t st);
}

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
int gfs_set_error(int code, const std::string &msg) { return fail(code, msg); }      // for multi.hip
#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(GFS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
    } while (0)

// ---- host restatements (bit-exact; this TU is built with -ffp-contract=off) ----------------
static inline int32_t h_sat_i32(double v) {
    if (v != v) return 0;
    if (v <= -2147483648.0) return INT32_MIN;
    if (v >= 2147483647.0) return INT32_MAX;
    return (int32_t)v;
}
static double h_fpp(double a, double b) {                                  // sgd.rs:155-182
    int32_t e = h_sat_i32(b);
    uint64_t bits; std::memcpy(&bits, &a, 8);
    int32_t high = (int32_t)(bits >> 32);
    int32_t diff = (int32_t)((uint32_t)high - 1072632447u);
    int32_t new_high = h_sat_i32((b - (double)e) * (double)diff + 1072632447.0);
    uint64_t fb = ((uint64_t)(uint32_t)new_high) << 32;
    double frac; std::memcpy(&frac, &fb, 8);
    double base = a, r = 1.0;
    int32_t ex = e;
    if (ex < 0) return std::nan("");     // the reference would loop forever (b < 0 never occurs for theta in [0,1))
    while (ex != 0) { if (ex & 1) r *= base; base *= base; ex >>= 1; }
    return r * frac;
}
static uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int check_params(const gfs_sgd_params *p) {
    if (!p) return fail(GFS_E_ARG, "params is null");
    if (!(p->theta >= 0.0 && p->theta < 1.0)) return fail(GFS_E_ARG, "theta must be in [0,1)");
    if (p->space_quantization_step == 0) return fail(GFS_E_ARG, "space_quantization_step must be > 0");
    if (!(p->eta_max > 0.0)) return fail(GFS_E_ARG, "eta_max must be > 0");
    return GFS_OK;
}

extern "C" {

const char *gfs_version(void) { return "gfasort_hip 0.1.0 (gfx950)"; }
const char *gfs_last_error(void) { return g_err.c_str(); }
int gfs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gfs_warmup(int device) {
    // Creates the HIP context (the first HIP call of a process costs ~0.1-0.3 s); callers run this
    // on a side thread while they are still parsing their input.
    const bool timing = std::getenv("GFS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(GFS_E_HIP, "no HIP device available (libgfasort_hip has no CPU fallback)");
    if (device < 0 || device >= n) return fail(GFS_E_ARG, "bad device index");
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gfs_warmup] %-12s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(nullptr));
    lap("context");
    // code objects load on first use, one per translation unit: touch each now, and the allocator too
    HIPCHK(gfs::warm_module_1d());
    lap("module 1d");
    HIPCHK(gfs::warm_module_index());
    lap("module index");
    HIPCHK(gfs::warm_module_nd());
    HIPCHK(gfs::warm_module_nd_team());
    lap("modules nd");
    // ... and the copy paths in both directions (the first hipMemcpy of a process sets up its staging
    // buffers and DMA queues: ~0.13 s when it was left to the first upload)
    void *p = nullptr;
    std::vector<unsigned char> h(1 << 20, 0);
    HIPCHK(hipMalloc(&p, 64u << 20));
    lap("malloc");
    HIPCHK(hipMemcpy(p, h.data(), h.size(), hipMemcpyHostToDevice));
    lap("h2d");
    HIPCHK(hipMemset(p, 0, 64u << 20));
    HIPCHK(hipDeviceSynchronize());
    lap("memset");
    HIPCHK(hipMemcpy(h.data(), p, h.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipDeviceSynchronize());
    lap("d2h");
    HIPCHK(hipFree(p));
    lap("free");
    return GFS_OK;
}

double gfs_fast_precise_pow(double a, double b) { return h_fpp(a, b); }

int gfs_sgd_schedule(const gfs_sgd_params *p, double *etas) {              // sgd.rs:300-308,617-638
    if (!p || !etas) return fail(GFS_E_ARG, "null argument");
    double w_min = 1.0 / p->eta_max, w_max = 1.0;
    double eta_max = 1.0 / w_min;
    double eta_min = p->eps / w_max;
    double lambda = std::log(eta_max / eta_min) / ((double)p->iter_max - 1.0);
    for (uint64_t t = 0; t <= p->iter_max; ++t) {
        int64_t d = (int64_t)t - (int64_t)p->iter_with_max_learning_rate;
        if (d < 0) d = -d;
        etas[t] = eta_max * std::exp(-lambda * (double)d);
    }
    return GFS_OK;
}

uint64_t gfs_zeta_table_len(const gfs_sgd_params *p) {                     // sgd.rs:311-315
    if (!p || p->space_quantization_step == 0) return 0;
    uint64_t n = p->space <= p->space_max
                     ? p->space
                     : p->space_max + (p->space - p->space_max) / p->space_quantization_step + 1;
    return n + 1;
}

int gfs_zeta_table(const gfs_sgd_params *p, double *zetas) {               // sgd.rs:317-331
    if (!p || !zetas) return fail(GFS_E_ARG, "null argument");
    uint64_t len = gfs_zeta_table_len(p);
    if (!len) return fail(GFS_E_ARG, "bad zeta parameters");
    for (uint64_t k = 0; k < len; ++k) zetas[k] = 0.0;
    double zeta_tmp = 0.0;
    for (uint64_t i = 1; i <= p->space; ++i) {
        zeta_tmp += h_fpp(1.0 / (double)i, p->theta);
        if (i <= p->space_max) zetas[i] = zeta_tmp;
        if (i >= p->space_max && (i - p->space_max) % p->space_quantization_step == 0) {
            uint64_t idx = p->space_max + 1 + (i - p->space_max) / p->space_quantization_step;
            if (idx < len) zetas[idx] = zeta_tmp;
        }
    }
    return GFS_OK;
}

int gfs_init_positions(const gfs_graph_view *g, double *x) {               // sgd.rs:271-294
    if (!g || (!x && g->n_nodes)) return fail(GFS_E_ARG, "null argument");
    uint64_t len = 0;
    for (uint64_t i = 0; i < g->n_nodes; ++i) { x[i] = (double)len; len += g->node_len[i]; }
    return GFS_OK;
}

int gfs_init_layout_dim0(const gfs_graph_view *g, uint64_t D, double *c) { // sgd.rs:832-853
    if (!g || (!c && g->n_nodes) || D == 0) return fail(GFS_E_ARG, "bad argument");
    uint64_t len = 0;
    for (uint64_t i = 0; i < g->n_nodes; ++i) {
        c[i * 2 * D + 0] = (double)len;
        c[i * 2 * D + D] = (double)(len + g->node_len[i]);
        len += g->node_len[i];
    }
    return GFS_OK;
}

// rand_distr 0.5 StandardNormal (f64) on Xoshiro256+ — the 256-layer ziggurat, restated from the crate's published
// algorithm; its tables are rebuilt by the construction of the crate's generator script (R, V below).  PARITY UNPINNED
// (DESIGN.md §5): neither the crate nor its table literals are in the container.
namespace {
constexpr double kZigR = 3.6541528853610088, kZigV = 0.00492867323399;
struct ZigTables {
    double x[257], f[257];
    ZigTables() {
        x[0] = kZigV / std::exp(-kZigR * kZigR / 2.0);
        x[1] = kZigR;
        for (int i = 1; i < 256; ++i) x[i + 1] = std::sqrt(-2.0 * std::log(kZigV / x[i] + std::exp(-x[i] * x[i] / 2.0)));
        x[256] = 0.0;
        for (int i = 0; i <= 256; ++i) f[i] = std::exp(-x[i] * x[i] / 2.0);
    }
};
struct Xo256p {                                                            // rand_xoshiro 0.7 Xoshiro256Plus
    uint64_t s[4];
    explicit Xo256p(uint64_t seed) { for (auto &w : s) w = splitmix64(seed); }   // seed_from_u64
    uint64_t next() {
        const uint64_t r = s[0] + s[3], t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = (s[3] << 45) | (s[3] >> 19);
        return r;
    }
};
inline double float_with_exponent(uint64_t fraction52, int e) {
    const uint64_t b = fraction52 | ((uint64_t)(1023 + e) << 52);
    double d; std::memcpy(&d, &b, 8); return d;
}
inline double open01(Xo256p &g) { return float_with_exponent(g.next() >> 12, 0) - (1.0 - 2.220446049250313e-16 / 2.0); }
double standard_normal(Xo256p &g) {
    static const ZigTables T;
    for (;;) {
        const uint64_t bits = g.next();
        const unsigned i = (unsigned)(bits & 0xff);
        const double u = float_with_exponent(bits >> 12, 1) - 3.0;               // [-1, 1)
        const double x = u * T.x[i];
        if (std::fabs(x) < T.x[i + 1]) return x;
        if (i == 0) {                                                          // the tail beyond R
            double tx = 1.0, ty = 0.0;
            while (-2.0 * ty < tx * tx) {
                const double x_ = open01(g), y_ = open01(g);
                tx = std::log(x_) / kZigR; ty = std::log(y_);
            }
            return u < 0.0 ? tx - kZigR : kZigR - tx;
        }
        const double r = (double)(g.next() >> 11) * (1.0 / 9007199254740992.0);  // rng.random::<f64>()
        if (T.f[i + 1] + (T.f[i] - T.f[i + 1]) * r < std::exp(-x * x / 2.0)) return x;
    }
}
}  // namespace

// The whole start of path_linear_sgd_layout (sgd.rs:829-853): one generator seeded `seed`; per node the + end's
// dimensions 1..D-1, then the - end's, each StandardNormal * sqrt(2N); dimension 0 as gfs_init_layout_dim0.
int gfs_init_layout(const gfs_graph_view *g, uint64_t D, uint64_t seed, double *c) {
    if (!g || (!c && g->n_nodes) || D == 0) return fail(GFS_E_ARG, "bad argument");
    Xo256p rng(seed);                                                          // :829
    const double sqrt_n = std::sqrt((double)g->n_nodes * 2.0);                 // :836
    uint64_t len = 0;
    for (uint64_t i = 0; i < g->n_nodes; ++i) {
        c[i * 2 * D + 0] = (double)len;                                        // :839
        for (uint64_t d = 1; d < D; ++d) c[i * 2 * D + d] = standard_normal(rng) * sqrt_n;          // :840-843
        c[i * 2 * D + D] = (double)(len + g->node_len[i]);                     // :846
        for (uint64_t d = 1; d < D; ++d) c[i * 2 * D + D + d] = standard_normal(rng) * sqrt_n;      // :847-850
        len += g->node_len[i];
    }
    return GFS_OK;
}

int gfs_sort_order(const double *x, uint64_t n, uint64_t *order) {         // sgd.rs:665-671
    if ((!x || !order) && n) return fail(GFS_E_ARG, "null argument");
    std::iota(order, order + n, (uint64_t)0);
    // partial_cmp(..).unwrap_or(Equal) + stable sort: ascending, -0.0 == +0.0, ties keep the index
    // order.  NaNs (never produced by a finite run) are placed after all numbers so that the order
    // is total; the device version (gfs_ctx_sort_order) uses the same rule.
    std::stable_sort(order, order + n, [x](uint64_t a, uint64_t b) {
        const double xa = x[a], xb = x[b];
        if (xa != xa) return false;
        if (xb != xb) return true;
        return xa < xb;
    });
    return GFS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// resident context
// ---------------------------------------------------------------------------------------------
static constexpr size_t kCounterBytes = 1024 * 8 * sizeof(unsigned long long);   // gfs::COUNTER_SLOTS lines of 64 B

struct gfs_ctx {
    int device = 0;
    int cu_count = 0;
    uint64_t n_nodes = 0, n_steps = 0, n_paths = 0;
    uint32_t max_path_steps = 0;
    bool valid_paths = false;
    std::vector<uint32_t> path_counts;
    std::vector<uint32_t> perm;        // dense node index (ABI order) -> internal index (device layout of positions)
    // device mirror of PathIndex
    uint4 *d_step_rec = nullptr;
    uint4 *d_path_rec = nullptr;
    uint64_t *d_path_len = nullptr;
    uint32_t *d_perm = nullptr;         // node layout on the device (dense index -> slot)
    uint32_t *d_node_len = nullptr;     // node lengths by dense index (K4: initial positions)
    // SGD state
    int dims = 0;                      // 0 = 1D
    bool configured = false;
    gfs_sgd_params params{};
    gfs_launch_config cfg{};
    std::vector<double> etas;
    double *d_zetas = nullptr; uint64_t zlen_full = 0, zlen_staged = 0;
    double *d_x = nullptr; bool x_owned = false; uint64_t x_len = 0;
    uint64_t *d_rng = nullptr;
    uint32_t *d_lead = nullptr;      // team kernels: the waves' partly expanded passes, [8][n_streams]
    unsigned long long *d_counters = nullptr;
    gfs_term *d_trace = nullptr; uint32_t *d_trace_cnt = nullptr;
    gfs::IterConsts *d_its = nullptr; uint64_t its_cap = 0;   // schedule slice of a fused launch (arbitrary lists)
    gfs::IterConsts *d_its_all = nullptr;                     // constants of iterations 0..=iter_max, resident
    uint32_t *d_pool = nullptr; uint64_t pool_cap = 0;        // fused launch: per-iteration work pool counters (sgd_kernels_1d.hip)
    uint64_t n_streams = 0, quota_total = 0;
    uint64_t fused_resident_blocks = 0; // workgroups of the fused 1D kernel the chip holds at once (block size, LDS table)
    uint32_t block = 256;
    uint32_t bundle = 1;               // lanes per sampling bundle actually used (1 = reference streams)
    uint32_t partners = 1;             // partner draws per leader (2: 1D team kernel at B = 64)
    uint32_t chain = 1;                // longest run in trips (sgd_device.h run_trips); 1 = a run is one trip
    bool lds_tables = true, atomic_loads = true;
    size_t lds_bytes = 0;
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double kernel_ms_harvested = 0.0;  // durations of event pairs already recycled
    uint64_t iterations = 0;
    uint64_t launches = 0;
    double total_ms = 0.0;
};

static void free_sgd_state(gfs_ctx *c) {
    if (c->d_zetas) (void)hipFree(c->d_zetas);
    if (c->d_x && c->x_owned) (void)hipFree(c->d_x);
    if (c->d_rng) (void)hipFree(c->d_rng);
    if (c->d_lead) (void)hipFree(c->d_lead);
    c->d_lead = nullptr;
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_trace) (void)hipFree(c->d_trace);
    if (c->d_trace_cnt) (void)hipFree(c->d_trace_cnt);
    if (c->d_its) (void)hipFree(c->d_its);
    if (c->d_its_all) (void)hipFree(c->d_its_all);
    c->d_its = nullptr; c->its_cap = 0; c->d_its_all = nullptr;
    if (c->d_pool) (void)hipFree(c->d_pool);
    c->d_pool = nullptr; c->pool_cap = 0;
    c->d_zetas = nullptr; c->d_x = nullptr; c->d_rng = nullptr; c->d_counters = nullptr;
    c->d_trace = nullptr; c->d_trace_cnt = nullptr; c->x_owned = false; c->configured = false;
}

static int seed_streams(gfs_ctx *c) {
    // stream t <- Xoshiro256Plus::seed_from_u64(seed + stream_base + t)   (sgd.rs:431-432)
    const uint64_t T = c->n_streams;
    std::vector<uint64_t> st(4 * T);
    for (uint64_t t = 0; t < T; ++t) {
        uint64_t sm = c->params.seed + c->cfg.stream_base + t;
        for (int k = 0; k < 4; ++k) st[(uint64_t)k * T + t] = splitmix64(sm);
    }
    HIPCHK(hipMemcpy(c->d_rng, st.data(), st.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(c->d_counters, 0, kCounterBytes));
    if (c->d_trace_cnt) HIPCHK(hipMemset(c->d_trace_cnt, 0, T * sizeof(uint32_t)));
    c->events_used = 0; c->kernel_ms_harvested = 0.0; c->iterations = 0; c->total_ms = 0.0;
    return GFS_OK;
}

// The zeta table of sgd.rs:311-331 on the device.  Only indices reachable from
// jump <= min(space, max_path_steps-1) are ever read (sgd.rs:462-469); when the table is computed here
// the running sum — which is order-dependent and must be accumulated exactly as the reference does —
// stops at the largest i that feeds a reachable entry (space is the longest path in bp: 1.3e6 for C3,
// of which 1.6e5 matter).
static int upload_zeta_table(gfs_ctx *c, const gfs_sgd_params *p, const double *zetas) {
    c->zlen_full = gfs_zeta_table_len(p);
    if (c->zlen_full > 0xFFFFFFFFull) return fail(GFS_E_UNSUPPORTED, "zeta table too long");
    const uint64_t maxjump = std::min<uint64_t>(p->space, c->max_path_steps ? c->max_path_steps - 1 : 0);
    const uint64_t last_idx = maxjump > p->space_max
                                  ? p->space_max + (maxjump - p->space_max) / p->space_quantization_step + 1
                                  : maxjump;
    c->zlen_staged = std::min<uint64_t>(last_idx + 1, c->zlen_full);
    std::vector<double> ztab;
    if (!zetas) {
        const uint64_t m = c->zlen_staged - 1;                      // last staged index
        const uint64_t need_i = m <= p->space_max ? m : p->space_max + (m - p->space_max - 1) * p->space_quantization_step;
        gfs_sgd_params q = *p;
        q.space = std::min<uint64_t>(p->space, std::max<uint64_t>(need_i, 1));
        std::vector<double> part(gfs_zeta_table_len(&q));
        gfs_zeta_table(&q, part.data());
        ztab.assign(c->zlen_full, 0.0);
        std::copy(part.begin(), part.begin() + std::min<size_t>(part.size(), ztab.size()), ztab.begin());
        zetas = ztab.data();
    }
    HIPCHK(hipMalloc(&c->d_zetas, c->zlen_full * 8));
    HIPCHK(hipMemcpy(c->d_zetas, zetas, c->zlen_full * 8, hipMemcpyHostToDevice));
    return GFS_OK;
}

// Streams per launch when the caller leaves it to the library.
static uint64_t auto_stream_count(const gfs_ctx *c, bool team) {
    // Lanes per CU: each wave is a serial chain of memory round trips, so more chains raise throughput until the memory-side
    // atomic units saturate.  Round 1 (profiles/r01/sweep_streams_final.log, defer_probe.log): C3 69.1 / 78.7 / 80.2 G
    // updates/s at 512 / 768 / 976 lanes per CU, C4 layout flat from 768 up; reference streams flat within 5 % from 512 up.
    // Round 2: the 1D team kernels run 4 waves per SIMD (128 VGPRs; twin trips keep three blocks of a trip in flight) = 1024
    // lanes per CU; 5 waves (96 VGPRs) spill 58 registers and are slower (profiles/r02/two_partners.log).  The fused launch
    // further bounds the count by the workgroups that are resident at once (setup_common).
    // The layout team kernels live on registers (a twin trip holds six records and three ends' coordinates): built for 3 waves
    // per SIMD (165 VGPRs at D = 2, nothing spilled) = 768 lanes per CU, for 2 at D = 3 (176) = 512 (sgd_kernels_nd_team.hip
    // nd_team_waves; round 2's kernel needed 203 and ran two).
    const uint64_t chip = (uint64_t)c->cu_count * ((team && c->dims == 0) ? 1024 : (team && c->dims >= 2) ? 256u * (unsigned)gfs::nd_team_waves(c->dims) : 976);
    // keep >= 8 updates per stream per batch on small graphs
    const uint64_t by_work = ((c->quota_total + 7) / 8 + 63) / 64 * 64;
    // and never more than one stream per 4 nodes (<= 0.5 in-flight terms per node): every in-flight
    // term corrects its two nodes from positions read before the others landed, so with ~2 concurrent
    // terms per node and mu clamped at 1 the corrections overshoot — a 6000-node graph of short paths
    // diverged (stress 1e8) under 6784 reference streams and converges under 1024
    // (profiles/r01/stream_cap_probe.log).  The team kernels tolerate three streams per 4 nodes: with the work pools of the
    // fused launch, bubble graphs of 26k / 79k / 197k nodes keep their relative error at path distance 1 (0.194 / 0.206 /
    // 0.192-0.197 against 0.198 / 0.208 / 0.191 at one stream per 2 nodes; reference streams 0.192 / 0.201 / 0.190) up to
    // one stream per node and lose it at two (0.224 / 0.248 / 0.220), at 2.0 / 1.65 / 1.2 times the rate
    // (profiles/r02/stream_cap_pools.log; round 1 allowed one per 2 nodes, measured with free-running waves whose drift
    // cost precision by itself).  An explicit n_streams overrides this.
    // Round 3 re-measured the bound (medium graphs leave the chip partly empty under it).  Bubble graphs of 66k / 131k / 302k
    // nodes keep every octave of the relative error within 4 % of reference streams up to 1.5 streams per node and lose distance
    // 1 at 2.0 (profiles/r03/stream_cap_probe.log) — but a window graph whose 16 paths each cover 5/8 of its 200k nodes loses its
    // exact chain order at 1.0 per node (3-115 inversions) and is scrambled at 1.25 (profiles/r03/chain_cap_probe.log), where
    // three per 4 nodes is exact on every graph tried.  The bound stays.
    const uint64_t by_nodes = (team ? c->n_nodes * 3 / 4 : c->n_nodes / 4) / 64 * 64;
    return std::max<uint64_t>(64, std::min(chip, std::min(by_work, by_nodes)));
}

// Sampling bundle: flags bits 16..23: 0 = auto, 1 = reference streams, 4..64 explicit (sgd_device.h).
static int choose_bundle(gfs_ctx *c, int dims) {
    const uint64_t T = c->n_streams;
    uint32_t b = (c->cfg.flags >> 16) & 0xFFu;
    const bool b_auto = b == 0;
    if (b > 1 && (T % 64 != 0 || (b != 4 && b != 8 && b != 16 && b != 32 && b != 64)))
        return fail(GFS_E_ARG, "bundled sampling needs n_streams % 64 == 0 and a bundle of 4, 8, 16, 32 or 64");
    if (b > 1 && dims != 0 && (dims > 3 || b == 4))
        return fail(GFS_E_UNSUPPORTED, "bundled layout kernels exist for 1..3 dimensions and bundles of 8..64");
    if (b == 0) {
        // auto (measured: profiles/r03/policy_sweep.log — bubble graphs of 16k...300k nodes, three seeds per cell, the relative
        // error per octave of path distance against reference streams): on graphs of >= 16 384 nodes the widest bundle for
        // which >= 95 % of the steps lie in paths of at least 4*B steps.  B = 64 with long runs is within 2-7 % of reference
        // streams in every octave from 16k nodes up and 2-10 times faster; narrower bundles and runs of one trip are both slower
        // and worse (+16...42 % at 64-127 steps from 131k nodes up: a run's two blocks move rigidly and leave a step at their edges,
        // short runs have more edges).  Round 2's extra condition — ">= 4096 independent leader draws per iteration" — is gone:
        // graphs with 37-99 leader draws per iteration are in that table and are as good as those with thousands; the run
        // length, not the number of leaders, is what the quality follows (bounded below by a floor of 64 leaders, see K).
        // Smaller graphs run reference streams: DRB1 (5k nodes) converged visibly slower with bundles (round 1).
        b = 1;
        if (T % 64 == 0 && dims <= 3 && c->n_nodes >= 16384) {
            for (uint32_t cand : {64u, 32u, 16u, 8u, 4u}) {
                if (cand == 4u && dims != 0) continue;
                uint64_t long_steps = 0;
                for (uint32_t cnt : c->path_counts) if (cnt >= 4 * cand) long_steps += cnt;
                if ((double)long_steps >= 0.95 * (double)c->n_steps) { b = cand; break; }
            }
        }
    }
    c->bundle = b;
    // Long runs (sgd_device.h run_trips): flags bits 24..31, 0 = auto.  Only the team kernels at B = 64 chain trips;
    // auto = 64 trips (runs of up to 4096 steps, adapted per path): the relative error of the layout, measured per octave
    // of path distance, is then within 10 % of reference streams on bubble graphs of 0.5M and 2M nodes — below it at
    // short distances — for the oracle's mirror and on the GPU (profiles/r02/quality_probe_long_runs.log).
    uint32_t k = (c->cfg.flags >> 24) & 0xFFu;
    if (k > 64 || (k & (k - 1))) return fail(GFS_E_ARG, "GFS_F_CHAIN: the run length in trips must be a power of two <= 64");
    // (layout kernels: 16 — on C4 runs of 64 trips cost 13 % of the rate, 30.8 against 34.2-35.5 G updates/s, and the error
    // profile of the 2-D layout is already below reference streams' at 16: profiles/r02/quality_probe_layout_k.log)
    const bool k_auto = k == 0;
    if (k == 0) k = dims ? 16 : 64;
    // Two partners per leader (sgd_device.h Leader): the team kernels at B = 64 (1D; layouts of 2 and 3 dimensions), unless
    // GFS_F_ONE_PARTNER
    c->partners = (b == 64 && (dims == 0 || dims == 2 || dims == 3) && !(c->cfg.flags & GFS_F_ONE_PARTNER)) ? 2u : 1u;
    // ... auto: and short enough that an iteration still draws >= 64 leaders (a leader stands for up to 64 * K * partners
    // terms): at 16k nodes runs of 32 trips left 37 leaders per iteration and +6 % at path distance 1, runs of 16 (74 leaders)
    // +1 %; from 32k nodes up 37 leaders were within 3 % (same table).  Binds only below ~500k steps.
    // (only where the library picked the bundle as well: an explicit GFS_F_BUNDLE(64) keeps 64 / 16)
    if (k_auto && b_auto && b == 64) while (k > 1 && c->quota_total / (64ull * k * c->partners) < 64) k >>= 1;
    c->chain = b == 64 ? k : 1;                       // (nD: team kernels exist for D <= 3; checked above)
    return GFS_OK;
}

static void iter_consts(const gfs_ctx *c, uint64_t k, gfs::IterConsts &it);

static int setup_common(gfs_ctx *c, const gfs_sgd_params *p, int dims, const gfs_launch_config *cfg,
                        const double *etas, const double *zetas) {
    if (!c) return fail(GFS_E_ARG, "ctx is null");
    int rc = check_params(p);
    if (rc) return rc;
    if (dims < 0 || dims > GFS_MAX_DIMS) return fail(GFS_E_UNSUPPORTED, "dimensions must be 1..8");
    HIPCHK(hipSetDevice(c->device));
    free_sgd_state(c);
    c->params = *p;
    c->cfg = cfg ? *cfg : gfs_launch_config{};
    c->dims = dims;
    c->x_len = 0;
    if (c->n_nodes == 0) { c->configured = true; return GFS_NOTHING_TO_DO; }

    // positions.  A context with nothing to do (no path of more than one step: sgd.rs:250-261 returns before any
    // update) still owns a full-length position replica: a multi-GPU rank whose shard holds no such path must be able to
    // upload, bind, merge and download like its peers — its contribution to every merge is a zero delta of the same size.
    c->x_len = dims ? c->n_nodes * 2 * (uint64_t)dims : c->n_nodes;
    HIPCHK(hipMalloc(&c->d_x, c->x_len * 8));
    HIPCHK(hipMemset(c->d_x, 0, c->x_len * 8));
    c->x_owned = true;
    if (!c->valid_paths) { c->configured = true; return GFS_NOTHING_TO_DO; }

    // eta schedule and zeta table (host, bit-exact) unless supplied
    c->etas.resize(p->iter_max + 1);
    if (etas) std::copy(etas, etas + p->iter_max + 1, c->etas.begin());
    else gfs_sgd_schedule(p, c->etas.data());
    rc = upload_zeta_table(c, p, zetas);
    if (rc) return rc;

    // launch shape
    c->quota_total = c->cfg.term_updates_per_iteration ? c->cfg.term_updates_per_iteration : p->min_term_updates;
    c->block = c->cfg.block_size ? c->cfg.block_size : 256;
    if (c->block % 64 || c->block > 1024) return fail(GFS_E_ARG, "block_size must be a multiple of 64, <= 1024");
    uint64_t T = c->cfg.n_streams ? c->cfg.n_streams : auto_stream_count(c, false);
    if (T > 0x7FFFFFFFull) return fail(GFS_E_ARG, "n_streams too large");
    c->n_streams = T;
    if (c->quota_total / T + 1 > 0xFFFFFFFFull) return fail(GFS_E_UNSUPPORTED, "per-stream quota exceeds 2^32");
    if (c->cfg.attempt_factor == 0) c->cfg.attempt_factor = 64;
    if (c->cfg.attempt_factor > 0xFFFFFFFFull) return fail(GFS_E_ARG, "attempt_factor too large");
    rc = choose_bundle(c, dims);
    if (rc) return rc;
    if (!c->cfg.n_streams && c->bundle > 1) c->n_streams = T = auto_stream_count(c, true);   // both counts are multiples of 64
    if (dims != 0 && c->bundle > 1 && c->block > 256)
        return fail(GFS_E_ARG, "the layout team kernels are built for workgroups of at most 256 lanes");
    c->atomic_loads = !(c->cfg.flags & GFS_F_PLAIN_LOADS);
    size_t lds = (size_t)c->n_paths * sizeof(uint4) + (size_t)c->zlen_staged * 8;
    c->lds_tables = !(c->cfg.flags & GFS_F_NO_LDS_TABLES) && lds <= 48 * 1024;
    c->lds_bytes = c->lds_tables ? lds : 0;
    c->fused_resident_blocks = 0;
    if (dims == 0 && c->bundle >= 16) {
        // The fused launch has no grid barrier: a workgroup that does not fit on the chip beside the others would walk
        // its whole schedule, early large-eta iterations included, after they have finished theirs — on a 525k-node graph
        // 5 such waves of 4101 were enough to wreck the layout (relative error 64 at path distance 1:
        // profiles/r02/streams_5_waves.log).  So the fused kernel is only launched with every workgroup resident: ask the
        // runtime how many fit per CU with this block size and LDS table (33 KB of zeta table = 4 blocks of 256 per CU, not
        // 5), bound the automatic stream count by it, and run one launch per iteration when a caller asks for more streams.
        int per_cu = 0;
        HIPCHK(gfs::prepare_1d_fused(c->bundle, c->lds_tables, (int)c->block, c->lds_bytes, &per_cu));   // (also: not inside the first launch's event bracket)
        c->fused_resident_blocks = (uint64_t)std::max(per_cu, 0) * c->cu_count;
        const uint64_t resident = c->fused_resident_blocks * c->block;
        if (!c->cfg.n_streams && T > resident && resident >= 64) c->n_streams = T = resident;
    }
    if ((dims == 2 || dims == 3) && c->bundle == 64) {
        // the layout team kernel's fused launch (K2c): the same residency rule
        int per_cu = 0;
        HIPCHK(gfs::prepare_nd_team_fused(dims, c->bundle, c->lds_tables, (int)c->block, c->lds_bytes, &per_cu));
        c->fused_resident_blocks = (uint64_t)std::max(per_cu, 0) * c->cu_count;
        const uint64_t resident = c->fused_resident_blocks * c->block;
        if (!c->cfg.n_streams && T > resident && resident >= 64) c->n_streams = T = resident;
    }

    HIPCHK(hipMalloc(&c->d_rng, 4 * T * 8));
    if (c->bundle > 1) {                                                 // team kernels, sort and layout
        HIPCHK(hipMalloc(&c->d_lead, 8 * T * sizeof(uint32_t)));
        HIPCHK(hipMemset(c->d_lead, 0, 8 * T * sizeof(uint32_t)));      // trips left = 0: no pass yet
    }
    HIPCHK(hipMalloc(&c->d_counters, kCounterBytes));
    if (c->cfg.trace_per_stream) {
        HIPCHK(hipMalloc(&c->d_trace, T * c->cfg.trace_per_stream * sizeof(gfs_term)));
        HIPCHK(hipMemset(c->d_trace, 0, T * c->cfg.trace_per_stream * sizeof(gfs_term)));
        HIPCHK(hipMalloc(&c->d_trace_cnt, T * sizeof(uint32_t)));
    }
    rc = seed_streams(c);
    if (rc) return rc;
    if (((dims == 0 && c->bundle >= 16) || ((dims == 2 || dims == 3) && c->bundle == 64) || c->bundle == 1) && c->params.iter_max < (1u << 20)) {
        // the whole schedule's per-iteration constants, for fused launches over consecutive iterations
        std::vector<gfs::IterConsts> all(c->params.iter_max + 1);
        for (uint64_t k = 0; k <= c->params.iter_max; ++k) iter_consts(c, k, all[k]);
        HIPCHK(hipMalloc(&c->d_its_all, all.size() * sizeof(gfs::IterConsts)));
        HIPCHK(hipMemcpy(c->d_its_all, all.data(), all.size() * sizeof(gfs::IterConsts), hipMemcpyHostToDevice));
    }
    c->configured = true;
    return GFS_OK;
}

static void iter_consts(const gfs_ctx *c, uint64_t k, gfs::IterConsts &it) {
    const gfs_sgd_params &p = c->params;
    double fc = std::floor(p.cooling_start * (double)p.iter_max);          // sgd.rs:297
    uint64_t first_cooling = !(fc > 0.0) ? 0 : (fc >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)fc);
    bool cooling = k > first_cooling;                                      // sgd.rs:393-396
    double theta = cooling ? 0.001 : p.theta;
    it.eta = c->etas[k];
    it.cooling = cooling ? 1 : 0;
    it.zeta2theta = 1.0 + h_fpp(0.5, theta);                               // sgd.rs:471 (== :143 bound)
    double omt = 1.0 - theta;                                              // sgd.rs:133
    it.omt_e = h_sat_i32(omt); it.omt_fb = omt - (double)it.omt_e;
    double alpha = 1.0 / (1.0 - theta);                                    // sgd.rs:132
    it.alpha_e = h_sat_i32(alpha); it.alpha_fb = alpha - (double)it.alpha_e;
    it._pad = 0;
}

extern "C" {

int gfs_ctx_create(const gfs_graph_view *g, int device, gfs_ctx **out) {
    return gfs_ctx_create_with_layout(g, device, nullptr, out);
}

int gfs_ctx_create_with_layout(const gfs_graph_view *g, int device, const uint32_t *node_perm, gfs_ctx **out) {
    if (!g || !out) return fail(GFS_E_ARG, "null argument");
    *out = nullptr;
    if (g->n_steps > (1ull << 40)) return fail(GFS_E_UNSUPPORTED, "more than 2^40 path steps");
    if (g->n_nodes > 0x7FFFFFFFull) return fail(GFS_E_UNSUPPORTED, "more than 2^31-1 nodes");
    if (g->n_paths > 0x7FFFFFFFull) return fail(GFS_E_UNSUPPORTED, "more than 2^31-1 paths");
    if (g->n_steps && (!g->step_node || !g->step_is_rev)) return fail(GFS_E_ARG, "null step arrays");
    if (!g->path_first_step) return fail(GFS_E_ARG, "null path_first_step");
    if (g->n_nodes && !g->node_len) return fail(GFS_E_ARG, "null node_len");
    if (g->path_first_step[0] != 0 || g->path_first_step[g->n_paths] != g->n_steps)
        return fail(GFS_E_ARG, "path_first_step must start at 0 and end at n_steps");
    for (uint64_t p = 0; p < g->n_paths; ++p)
        if (g->path_first_step[p + 1] < g->path_first_step[p]) return fail(GFS_E_ARG, "path_first_step not monotone");
    // (step_node's range is checked on the device, in the pass that derives the node layout)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(GFS_E_HIP, "no HIP device available (libgfasort_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GFS_E_ARG, "bad device index");
    const bool timing = std::getenv("GFS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gfs_ctx_create] %-12s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    lap("validate");
    gfs_ctx *c = new (std::nothrow) gfs_ctx();
    if (!c) return fail(GFS_E_NOMEM, "out of memory");
    c->device = device;
    c->n_nodes = g->n_nodes; c->n_steps = g->n_steps; c->n_paths = g->n_paths;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        delete c; return fail(GFS_E_HIP, "hipGetDeviceProperties failed");
    }
    c->cu_count = prop.multiProcessorCount;

    // Internal node layout.  The position vector is stored in FIRST-VISIT PATH ORDER (nodes in the
    // order the paths first step on them, unvisited nodes last; derived on the device, index_kernels.hip)
    // unless the caller supplies a layout: consecutive steps of a path then touch neighbouring position words whatever the
    // order of the input's S lines was, which is what lets a run's loads and atomics coalesce
    // (C3 with randomly ordered nodes: 63 G updates/s in path order, 11 G/s in input order).
    c->perm.assign(g->n_nodes, 0xFFFFFFFFu);
    if (node_perm) {
        std::vector<uint8_t> seen(g->n_nodes, 0);
        for (uint64_t k = 0; k < g->n_nodes; ++k) {
            if (node_perm[k] >= g->n_nodes || seen[node_perm[k]]) { delete c; return fail(GFS_E_ARG, "node_perm is not a permutation"); }
            seen[node_perm[k]] = 1; c->perm[k] = node_perm[k];
        }
    }
    // Path records (host, P entries) and the facts the launch logic needs
    std::vector<uint4> prec(std::max<uint64_t>(g->n_paths, 1));
    for (uint64_t p = 0; p < g->n_paths; ++p) {
        uint64_t b = g->path_first_step[p], e = g->path_first_step[p + 1];
        uint32_t cnt = (uint32_t)(e - b);
        if (e - b > 0xFFFFFFFFull) { delete c; return fail(GFS_E_UNSUPPORTED, "a path with more than 2^32-1 steps"); }
        prec[p].x = (uint32_t)b; prec[p].y = cnt;
        prec[p].z = cnt ? (uint32_t)(0u - cnt) % cnt : 0u; prec[p].w = (uint32_t)(b >> 32);
        c->path_counts.push_back(cnt);
        if (cnt > 1) c->valid_paths = true;                               // sgd.rs:250-256
        c->max_path_steps = std::max(c->max_path_steps, cnt);
    }
    {   // a step record keeps 55 bits of a step's bp position (its top bits carry the crowding exponents): refuse longer paths
        // instead of truncating them.  Cheap bound first (steps x longest node), the exact sum only where that fails.
        uint32_t max_len = 0;
        for (uint64_t k = 0; k < g->n_nodes; ++k) max_len = std::max(max_len, g->node_len[k]);
        for (uint64_t p = 0; p < g->n_paths; ++p) {
            const uint64_t b = g->path_first_step[p], e2 = g->path_first_step[p + 1];
            if ((unsigned __int128)(e2 - b) * max_len < ((unsigned __int128)1 << 55)) continue;
            unsigned __int128 bp = 0;
            for (uint64_t s2 = b; s2 < e2; ++s2) { const uint32_t n = g->step_node[s2]; if (n < g->n_nodes) bp += g->node_len[n]; }
            if (bp >= ((unsigned __int128)1 << 55)) { delete c; return fail(GFS_E_UNSUPPORTED, "a path of 2^55 bp or more"); }
        }
    }
    // K3 on the device: PathIndex::from_graph (sgd.rs:34-71) — the per-path exclusive prefix sum of
    // node lengths over the steps — written straight into the 16-byte step records
    // (index_kernels.hip).  Uploads 5 B per step instead of 16.
    auto bail = [&](const char *what, hipError_t e) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        gfs_ctx_destroy(c);
        return fail(GFS_E_HIP, m);
    };
    const uint64_t S = g->n_steps, N = g->n_nodes, P = g->n_paths;
    uint32_t *d_step_node = nullptr; uint8_t *d_rev = nullptr; uint64_t *d_first = nullptr, *d_tmp = nullptr;
    auto free_tmp = [&]() {
        if (d_step_node) (void)hipFree(d_step_node);
        if (d_rev) (void)hipFree(d_rev); if (d_first) (void)hipFree(d_first); if (d_tmp) (void)hipFree(d_tmp);
    };
    hipError_t e;
#define GFS_TRY(what, expr) if ((e = (expr)) != hipSuccess) { free_tmp(); return bail(what, e); }
    // (one record of padding, zeroed: the layout team kernels read the record AFTER a step for its node's length and use it
    // only where that step is not its path's last — so the graph's very last step needs no clamp, sgd_kernels_nd_team.hip)
    GFS_TRY("hipMalloc step_rec", hipMalloc(&c->d_step_rec, (S + 1) * sizeof(uint4)));
    GFS_TRY("hipMemset step_rec", hipMemset(c->d_step_rec + S, 0, sizeof(uint4)));
    GFS_TRY("hipMalloc path_rec", hipMalloc(&c->d_path_rec, prec.size() * sizeof(uint4)));
    GFS_TRY("hipMalloc path_len", hipMalloc(&c->d_path_len, std::max<uint64_t>(P, 1) * 8));
    GFS_TRY("hipMalloc perm", hipMalloc(&c->d_perm, std::max<uint64_t>(N, 1) * 4));
    GFS_TRY("hipMalloc node_len", hipMalloc(&c->d_node_len, std::max<uint64_t>(N, 1) * 4));
    if (N) { GFS_TRY("hipMemcpy node_len", hipMemcpy(c->d_node_len, g->node_len, N * 4, hipMemcpyHostToDevice)); }
    GFS_TRY("hipMemcpy path_rec", hipMemcpy(c->d_path_rec, prec.data(), prec.size() * sizeof(uint4), hipMemcpyHostToDevice));
    if (S) {
        GFS_TRY("hipMalloc step_node", hipMalloc(&d_step_node, S * 4));
        GFS_TRY("hipMemcpy step_node", hipMemcpy(d_step_node, g->step_node, S * 4, hipMemcpyHostToDevice));
    }
    if (S) {
        GFS_TRY("hipMalloc path_first", hipMalloc(&d_first, (P + 1) * 8));
        GFS_TRY("hipMemcpy path_first", hipMemcpy(d_first, g->path_first_step, (P + 1) * 8, hipMemcpyHostToDevice));
    }
    lap("step upload");
    {
        // range check of step_node, and (unless the caller brought a layout) the first-visit order
        int bad = 0;
        uint32_t *d_scratch_perm = nullptr;
        uint32_t *target = c->d_perm;
        if (node_perm && N) { GFS_TRY("hipMalloc scratch", hipMalloc(&d_scratch_perm, N * 4)); target = d_scratch_perm; }
        e = gfs::first_visit_layout_device(d_step_node, S, N, d_first, (uint32_t)P, target, &bad);
        if (d_scratch_perm) (void)hipFree(d_scratch_perm);
        if (e != hipSuccess) { free_tmp(); return bail("first_visit_layout", e); }
        if (bad) { free_tmp(); gfs_ctx_destroy(c); return fail(GFS_E_ARG, "step_node out of range"); }
        if (N && node_perm) { GFS_TRY("hipMemcpy perm", hipMemcpy(c->d_perm, c->perm.data(), N * 4, hipMemcpyHostToDevice)); }
        if (N && !node_perm) { GFS_TRY("hipMemcpy perm", hipMemcpy(c->perm.data(), c->d_perm, N * 4, hipMemcpyDeviceToHost)); }
    }
    lap("node layout");
    if (S) {
        GFS_TRY("hipMalloc step_is_rev", hipMalloc(&d_rev, S));

        GFS_TRY("hipMalloc scan", hipMalloc(&d_tmp, (S + 1) * 8));
        lap("alloc tmp");
        GFS_TRY("hipMemcpy step_is_rev", hipMemcpy(d_rev, g->step_is_rev, S, hipMemcpyHostToDevice));

        lap("upload");
        GFS_TRY("build_path_index", gfs::build_path_index_device(d_step_node, d_rev, c->d_node_len, c->d_perm, d_first, (uint32_t)P, S, N,
                                                                  d_tmp, c->d_step_rec, c->d_path_len));
    }
#undef GFS_TRY
    lap("K3");
    free_tmp();
    lap("free tmp");
    *out = c;
    return GFS_OK;
}

void gfs_ctx_destroy(gfs_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_sgd_state(c);
    for (auto &ev : c->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (c->d_step_rec) (void)hipFree(c->d_step_rec);
    if (c->d_path_rec) (void)hipFree(c->d_path_rec);
    if (c->d_path_len) (void)hipFree(c->d_path_len);
    if (c->d_perm) (void)hipFree(c->d_perm);
    if (c->d_node_len) (void)hipFree(c->d_node_len);
    delete c;
}

int gfs_ctx_setup_1d(gfs_ctx *c, const gfs_sgd_params *p, const gfs_launch_config *cfg,
                     const double *etas, const double *zetas) {
    return setup_common(c, p, 0, cfg, etas, zetas);
}
int gfs_ctx_setup_nd(gfs_ctx *c, const gfs_layout_params *p, const gfs_launch_config *cfg,
                     const double *etas, const double *zetas) {
    if (!p) return fail(GFS_E_ARG, "params is null");
    if (p->dimensions < 1 || p->dimensions > GFS_MAX_DIMS) return fail(GFS_E_UNSUPPORTED, "dimensions must be 1..8");
    return setup_common(c, &p->sgd, (int)p->dimensions, cfg, etas, zetas);
}

uint64_t gfs_ctx_positions_len(const gfs_ctx *c) { return c ? c->x_len : 0; }

int gfs_ctx_upload_positions(gfs_ctx *c, const double *host, uint64_t n) {
    if (!c || !host) return fail(GFS_E_ARG, "null argument");
    if (!c->d_x) return fail(GFS_E_STATE, "context not set up");
    if (n != c->x_len) return fail(GFS_E_ARG, "positions length mismatch");
    HIPCHK(hipSetDevice(c->device));
    // device order: 1D x[slot]; nD the planes coords[end][dim][slot] — reordered on the device
    double *d_stage = nullptr;
    HIPCHK(hipMalloc(&d_stage, n * 8));
    hipError_t e = hipMemcpy(d_stage, host, n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = gfs::reorder_positions_device(d_stage, c->d_x, c->d_perm, c->n_nodes, (uint32_t)c->dims, 1, 0);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_stage);
    if (e != hipSuccess) return fail(GFS_E_HIP, std::string("upload_positions: ") + hipGetErrorString(e));
    return GFS_OK;
}
int gfs_ctx_init_positions(gfs_ctx *c) {                                  // K4, sgd.rs:286-294
    if (!c) return fail(GFS_E_ARG, "ctx is null");
    if (!c->d_x || c->dims != 0) return fail(GFS_E_STATE, "needs a 1D context that has been set up");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    hipError_t e = gfs::init_positions_device(c->d_node_len, c->d_perm, c->d_x, c->n_nodes);
    if (e != hipSuccess) return fail(GFS_E_HIP, std::string("init_positions: ") + hipGetErrorString(e));
    return GFS_OK;
}
int gfs_ctx_download_positions(gfs_ctx *c, double *host, uint64_t n) {
    if (!c || !host) return fail(GFS_E_ARG, "null argument");
    if (!c->d_x) return fail(GFS_E_STATE, "context not set up");
    if (n != c->x_len) return fail(GFS_E_ARG, "positions length mismatch");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    double *d_stage = nullptr;
    HIPCHK(hipMalloc(&d_stage, n * 8));
    hipError_t e = gfs::reorder_positions_device(c->d_x, d_stage, c->d_perm, c->n_nodes, (uint32_t)c->dims, 0, 0);
    if (e == hipSuccess) e = hipMemcpy(host, d_stage, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_stage);
    if (e != hipSuccess) return fail(GFS_E_HIP, std::string("download_positions: ") + hipGetErrorString(e));
    return GFS_OK;
}
int gfs_ctx_node_layout(const gfs_ctx *c, uint32_t *perm_out, uint64_t n) {
    if (!c || !perm_out) return fail(GFS_E_ARG, "null argument");
    if (n != c->n_nodes) return fail(GFS_E_ARG, "layout length mismatch");
    std::copy(c->perm.begin(), c->perm.end(), perm_out);
    return GFS_OK;
}
void *gfs_ctx_positions_device(gfs_ctx *c) { return c ? (void *)c->d_x : nullptr; }
int gfs_ctx_bind_positions(gfs_ctx *c, void *device_ptr) {
    if (!c || !device_ptr) return fail(GFS_E_ARG, "null argument");
    if (!c->configured || !c->x_len) return fail(GFS_E_STATE, "context not set up");
    HIPCHK(hipSetDevice(c->device));
    if (c->d_x && c->x_owned) HIPCHK(hipFree(c->d_x));
    c->d_x = (double *)device_ptr; c->x_owned = false;
    return GFS_OK;
}
int gfs_ctx_reset_streams(gfs_ctx *c) {
    if (c && c->configured && !c->valid_paths) return GFS_NOTHING_TO_DO;
    if (!c || !c->d_rng) return fail(GFS_E_STATE, "context not set up");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    if (c->d_lead) HIPCHK(hipMemset(c->d_lead, 0, 8 * c->n_streams * sizeof(uint32_t)));
    return seed_streams(c);
}

static void fill_kargs(const gfs_ctx *c, gfs::KArgs &a) {
    a.step_rec = c->d_step_rec; a.path_rec = c->d_path_rec; a.path_len = c->d_path_len;
    a.zetas = c->d_zetas; a.x = c->d_x; a.rng = c->d_rng; a.counters = c->d_counters;
    a.trace = c->d_trace; a.trace_cnt = c->d_trace_cnt; a.lead = c->d_lead;
    a.n_steps = c->n_steps;
    const bool wide = c->n_steps > 0xFFFFFFFFull || (c->cfg.flags & GFS_F_DBG_WIDE_INDEX);
    a.steps_thresh = wide ? (0ull - c->n_steps) % c->n_steps
                          : (uint64_t)((uint32_t)(0u - (uint32_t)c->n_steps) % (uint32_t)c->n_steps);
    a.n_paths = (uint32_t)c->n_paths;
    a.zlen_full = (uint32_t)c->zlen_full; a.zlen_staged = (uint32_t)c->zlen_staged;
    a.n_streams = (uint32_t)c->n_streams;
    a.quota_base = (uint32_t)(c->quota_total / c->n_streams);
    a.quota_rem = (uint32_t)(c->quota_total % c->n_streams);
    a.attempt_factor = (uint32_t)c->cfg.attempt_factor;
    a.trace_per_stream = (uint32_t)c->cfg.trace_per_stream;
    a.space = (uint32_t)std::min<uint64_t>(c->params.space, 0xFFFFFFFFull);
    a.space_max = (uint32_t)std::min<uint64_t>(c->params.space_max, 0xFFFFFFFFull);
    a.space_q = (uint32_t)std::min<uint64_t>(c->params.space_quantization_step, 0xFFFFFFFFull);
    a.dbg = (c->cfg.flags >> 8) & 0x7Fu;             // bit 0x40 = GFS_F_DBG_WIDE_INDEX >> 8
    if (c->cfg.flags & GFS_F_DBG_NO_FUSED_TRIP) a.dbg |= 0x100u;     // (GFS_F_DBG_NO_TWIN_TRIP 0x400 arrives as dbg bit 0x04)
    a.bundle = c->bundle;
    a.chain = c->chain;
    a.partners = c->partners;
    a.dbg2 = 0;
    if (const char *e = std::getenv("GFS_DBG2")) a.dbg2 = (uint32_t)std::atol(e);
    a.chunk = c->dims ? gfs::ND_TEAM_CHUNK : gfs::TEAM_CHUNK;
    a.ref_chunk = gfs::REF_CHUNK_PER_LANE;
    if (const char *e = std::getenv("GFS_DBG_REF_CHUNK")) { const long v = std::atol(e); if (v >= 1 && v <= 4096) a.ref_chunk = (uint32_t)v; }   // probe knob (scripts/ref_fused_probe.py)
    a.n_nodes = (uint32_t)c->n_nodes;
    {   // crowding onset (sgd_device.h crowd_shift): four times the concurrency of an average node
        const uint64_t per = c->n_steps / std::max<uint64_t>(2 * c->n_streams, 1);
        int lg = 0; while ((per >> (lg + 1)) != 0) ++lg;               // floor(log2(max(per, 1)))
        a.kshift = lg + 2;
    }
}

static int next_event_pair(gfs_ctx *c, std::pair<hipEvent_t, hipEvent_t> *&ev) {
    if (c->events_used == c->events.size()) {
        if (c->events.size() >= 4096) {
            // long-lived context: recycle the pool instead of growing it (one sync per 4096 launches)
            HIPCHK(hipEventSynchronize(c->events.back().second));
            for (auto &p : c->events) {
                float t = 0.f;
                if (hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) c->kernel_ms_harvested += t;
            }
            c->events_used = 0;
        } else {
            hipEvent_t e0, e1;
            HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
            c->events.emplace_back(e0, e1);
        }
    }
    ev = &c->events[c->events_used++];
    return GFS_OK;
}

int gfs_ctx_run_iteration(gfs_ctx *c, uint64_t k, void *hip_stream) {
    if (!c) return fail(GFS_E_ARG, "ctx is null");
    if (!c->configured) return fail(GFS_E_STATE, "context not set up");
    if (!c->valid_paths || c->n_nodes == 0) return GFS_NOTHING_TO_DO;
    if (k > c->params.iter_max) return fail(GFS_E_ARG, "iteration beyond iter_max");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)hip_stream;
    gfs::KArgs a{};
    fill_kargs(c, a);
    iter_consts(c, k, a.it);
    dim3 block(c->block), grid((unsigned)((c->n_streams + c->block - 1) / c->block));
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    int rc = next_event_pair(c, ev);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ev->first, st));
    hipError_t e = c->dims == 0
        ? gfs::launch_1d(a, c->lds_tables, c->atomic_loads, c->d_trace != nullptr, grid, block, c->lds_bytes, st)
        : gfs::launch_nd(c->dims, a, c->lds_tables, c->atomic_loads, c->d_trace != nullptr, grid, block, c->lds_bytes, st);
    if (e != hipSuccess) return fail(GFS_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    HIPCHK(hipEventRecord(ev->second, st));
    c->iterations++;
    c->launches++;
    return GFS_OK;
}

// A range of iterations ks[0..n) (each in 0..=iter_max): ONE fused launch for the team kernels (sgd1d_team_fused_kernel; layouts of 2
// and 3 dimensions: sgdnd_team_fused_kernel) and for reference streams (sgd1d_fused_kernel, sgdnd_fused_kernel); otherwise one launch per iteration.
int gfs_ctx_run_range(gfs_ctx *c, const uint64_t *ks, uint64_t n, void *hip_stream) {
    if (!c || (!ks && n)) return fail(GFS_E_ARG, "null argument");
    if (!c->configured) return fail(GFS_E_STATE, "context not set up");
    if (!c->valid_paths || c->n_nodes == 0) return GFS_NOTHING_TO_DO;
    for (uint64_t i = 0; i < n; ++i) if (ks[i] > c->params.iter_max) return fail(GFS_E_ARG, "iteration beyond iter_max");
    // One persistent launch for the range where a fused kernel exists: the 1D team kernel at its widest bundles (K1c) and
    // reference streams in any dimension (K1d / K2d).  The waves of a fused launch draw an iteration's updates from a work
    // pool (a share per counter beyond 2^31 — 3e10 updates per iteration — cannot be pooled: one launch per iteration
    // then, unless the diagnostic GFS_F_DBG_FREE_RUNNING asks for round 1's fixed quotas).  The team kernel is only fused
    // with every workgroup resident — which assumes this context has the device to itself: concurrent streams or a second
    // rank on the same device can delay a workgroup, harmlessly under pools (a late wave finds the counters exhausted and
    // leaves), not so with fixed quotas.
    const uint64_t n_waves = (c->n_streams + 63) / 64;
    // (layouts draw an iteration from ONE counter, sgd_kernels_nd_team.hip K2c: the whole iteration must stay below 2^31)
    const bool pool_ok = n_waves <= 0xFFFFFFFFull &&
                         c->quota_total / (c->dims != 0 && c->bundle > 1 ? 1u : gfs::pool_slots((uint32_t)n_waves)) < (1ull << 31);
    const bool free_running = (c->cfg.flags & GFS_F_DBG_FREE_RUNNING) != 0;
    const bool team_shape = (c->dims == 0 && c->bundle >= 16) || ((c->dims == 2 || c->dims == 3) && c->bundle == 64);   // K1c, K2c
    const bool team_fusable = team_shape && (pool_ok || free_running) &&
                              (c->n_streams + c->block - 1) / c->block <= c->fused_resident_blocks;   // every workgroup resident
    const bool ref_fusable = c->bundle == 1 && pool_ok;
    // A range of ONE layout iteration is drawn from the pool too where it is at least four chunks per wave: with fixed quotas a layout
    // launch's waves finish as far apart as their leaders' costs are (C4: 2.33 ms per iteration against 2.21 pooled, 2.04 inside
    // a fused range).  Not with shorter chunks for smaller iterations: the layout pool is ONE counter, and it takes ~2e7 claims/s
    // comfortably and 4e7 not (C4 in chunks of 1024 / 512 / 256: 2.41 / 3.05 / 5.24 ms).  Not for the sort either: its launches of
    // one iteration are short (C3: 0.16 ms with fixed quotas, 0.15 pooled in chunks of 1024, 0.10 inside a fused range)
    // (profiles/r03/one_iteration_launch_probe.log, launch_overhead_probe.log).
    uint32_t one_chunk = c->dims ? gfs::ND_TEAM_CHUNK : gfs::TEAM_CHUNK;
    const uint64_t per_wave = c->quota_total / n_waves;
    bool single_ok = team_fusable && pool_ok && c->dims != 0 && per_wave >= 4ull * one_chunk;
    if (const char *e = std::getenv("GFS_DBG_ONE_CHUNK")) {               // probe knob (scripts/one_iteration_launch_probe.py): also for the sort
        const long v = std::atol(e);
        if (v >= 64 && v <= 4096 && !(v & (v - 1))) { one_chunk = (uint32_t)v; single_ok = team_fusable && pool_ok; }
    }
    const bool can_fuse = (team_fusable || ref_fusable) && c->atomic_loads && !c->d_trace && (n > 1 || (n == 1 && single_ok)) &&
                          n <= 0xFFFFFFFFull && !(c->cfg.flags & GFS_F_NO_FUSE);
    if (!can_fuse) {
        for (uint64_t i = 0; i < n; ++i) { int rc = gfs_ctx_run_iteration(c, ks[i], hip_stream); if (rc) return rc; }
        return GFS_OK;
    }
    // a fused launch covers at most kMaxFusedIterations (its pool counters are 1 KB per iteration); longer ranges are
    // consecutive launches on the stream
    constexpr uint64_t kMaxFusedIterations = 4096;
    if (n > kMaxFusedIterations) {
        for (uint64_t off = 0; off < n; off += kMaxFusedIterations) {
            const uint64_t m = std::min(kMaxFusedIterations, n - off);
            int rc = m > 1 ? gfs_ctx_run_range(c, ks + off, m, hip_stream) : gfs_ctx_run_iteration(c, ks[off], hip_stream);
            if (rc) return rc;
        }
        return GFS_OK;
    }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)hip_stream;
    bool consecutive = c->d_its_all != nullptr;
    for (uint64_t i = 1; i < n && consecutive; ++i) consecutive = ks[i] == ks[0] + i;
    const gfs::IterConsts *d_slice = nullptr;
    if (consecutive) {
        d_slice = c->d_its_all + ks[0];               // resident table: nothing to upload, nothing to wait for
    } else {
        std::vector<gfs::IterConsts> its(n);
        for (uint64_t i = 0; i < n; ++i) iter_consts(c, ks[i], its[i]);
        if (c->its_cap < n) {
            if (c->d_its) HIPCHK(hipFree(c->d_its));
            c->d_its = nullptr; c->its_cap = 0;
            HIPCHK(hipMalloc(&c->d_its, n * sizeof(gfs::IterConsts)));
            c->its_cap = n;
        }
        HIPCHK(hipMemcpyAsync(c->d_its, its.data(), n * sizeof(gfs::IterConsts), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));             // `its` is a stack-lifetime staging buffer
        d_slice = c->d_its;
    }
    gfs::KArgs a{};
    fill_kargs(c, a);
    if (n == 1 && c->bundle > 1) a.chunk = one_chunk;
    if (const char *e = std::getenv("GFS_DBG_CHUNK")) {                    // probe knob: the chunk of every fused team launch
        const long v = std::atol(e);
        if (c->bundle > 1 && v >= 64 && v <= 16384 && !(v & (v - 1))) a.chunk = (uint32_t)v;
    }
    iter_consts(c, ks[0], a.it);
    dim3 block(c->block), grid((unsigned)((c->n_streams + c->block - 1) / c->block));
    // work pools (sgd_kernels_1d.hip): the waves draw an iteration's updates from shared counters, zeroed per launch
    uint32_t *pool = nullptr;
    if (pool_ok && !(free_running && c->bundle > 1)) {
        if (c->pool_cap < n) {
            if (c->d_pool) HIPCHK(hipFree(c->d_pool));
            c->d_pool = nullptr; c->pool_cap = 0;
            HIPCHK(hipMalloc(&c->d_pool, gfs::pool_bytes(n)));
            c->pool_cap = n;
        }
        HIPCHK(hipMemsetAsync(c->d_pool, 0, gfs::pool_bytes(n), st));
        pool = c->d_pool;
    }
    std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
    int rc = next_event_pair(c, ev);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ev->first, st));                // (the event pair brackets the kernel alone)
    hipError_t e = c->bundle > 1 ? (c->dims == 0 ? gfs::launch_1d_fused(a, d_slice, (uint32_t)n, c->lds_tables, pool, grid, block, c->lds_bytes, st)
                                                 : gfs::launch_nd_team_fused(c->dims, a, d_slice, (uint32_t)n, c->lds_tables, pool, grid, block, c->lds_bytes, st))
                   : c->dims == 0 ? gfs::launch_1d_ref_fused(a, d_slice, (uint32_t)n, c->lds_tables, pool, grid, block, c->lds_bytes, st)
                                  : gfs::launch_nd_ref_fused(c->dims, a, d_slice, (uint32_t)n, c->lds_tables, pool, grid, block, c->lds_bytes, st);
    if (e != hipSuccess) return fail(GFS_E_HIP, std::string("fused kernel launch: ") + hipGetErrorString(e));
    HIPCHK(hipEventRecord(ev->second, st));
    c->iterations += n;
    c->launches++;
    return GFS_OK;
}

int gfs_ctx_synchronize(gfs_ctx *c, void *hip_stream) {
    if (!c) return fail(GFS_E_ARG, "ctx is null");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
    return GFS_OK;
}

int gfs_ctx_run(gfs_ctx *c, void *hip_stream) {
    if (!c) return fail(GFS_E_ARG, "ctx is null");
    if (!c->configured) return fail(GFS_E_STATE, "context not set up");
    if (!c->valid_paths || c->n_nodes == 0) return GFS_NOTHING_TO_DO;
    auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<uint64_t> ks(c->params.iter_max + 1);                  // iter_max+1 batches (sgd.rs:383)
        std::iota(ks.begin(), ks.end(), (uint64_t)0);
        int rc = gfs_ctx_run_range(c, ks.data(), ks.size(), hip_stream);
        if (rc) return rc;
    }
    int rc = gfs_ctx_synchronize(c, hip_stream);
    if (rc) return rc;
    c->total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return GFS_OK;
}

int gfs_ctx_stats(gfs_ctx *c, gfs_stats *out) {
    if (!c || !out) return fail(GFS_E_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    if (!c->d_counters) return GFS_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    std::vector<unsigned long long> cnt(kCounterBytes / 8);
    HIPCHK(hipMemcpy(cnt.data(), c->d_counters, kCounterBytes, hipMemcpyDeviceToHost));
    for (size_t s = 0; s < cnt.size(); s += 8) { out->term_updates += cnt[s]; out->attempts += cnt[s + 1]; }
    out->iterations = c->iterations; out->n_streams = c->n_streams; out->bundle = c->bundle; out->run_trips = c->chain;
    double ms = c->kernel_ms_harvested;
    for (size_t k = 0; k < c->events_used; ++k) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, c->events[k].first, c->events[k].second) == hipSuccess) ms += t;
    }
    out->kernel_ms = ms; out->total_ms = c->total_ms; out->launches = c->launches;
    return GFS_OK;
}

int gfs_ctx_trace(gfs_ctx *c, gfs_term *out, uint64_t n_terms, uint64_t *counts, uint64_t n_streams) {
    if (!c || !out) return fail(GFS_E_ARG, "null argument");
    if (!c->d_trace) return fail(GFS_E_STATE, "trace_per_stream was 0");
    if (n_terms != c->n_streams * c->cfg.trace_per_stream) return fail(GFS_E_ARG, "trace length mismatch");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, c->d_trace, n_terms * sizeof(gfs_term), hipMemcpyDeviceToHost));
    {   // the kernels record internal node indices: translate to the ABI's dense indices
        std::vector<uint32_t> inv(c->n_nodes);
        for (uint64_t k = 0; k < c->n_nodes; ++k) inv[c->perm[k]] = (uint32_t)k;
        for (uint64_t t = 0; t < n_terms; ++t) {
            if (c->dims == 0) { if (out[t].d_ij != 0.0) { out[t].i = inv[out[t].i]; out[t].j = inv[out[t].j]; } }
            else if (out[t].d_ij != 0.0) { out[t].i = inv[out[t].i >> 1] * 2 + (out[t].i & 1); out[t].j = inv[out[t].j >> 1] * 2 + (out[t].j & 1); }
        }
    }
    if (counts) {
        if (n_streams != c->n_streams) return fail(GFS_E_ARG, "counts length mismatch");
        std::vector<uint32_t> tmp(c->n_streams);
        HIPCHK(hipMemcpy(tmp.data(), c->d_trace_cnt, tmp.size() * 4, hipMemcpyDeviceToHost));
        for (uint64_t t = 0; t < c->n_streams; ++t) counts[t] = tmp[t];
    }
    return GFS_OK;
}

// K6 on the device: rank order of the context's current positions (1D) — sgd.rs:665-671.
int gfs_ctx_sort_order(gfs_ctx *c, uint64_t *order, uint64_t n) {
    if (!c || (!order && n)) return fail(GFS_E_ARG, "null argument");
    if (!c->d_x || c->dims != 0) return fail(GFS_E_STATE, "needs a 1D context that has been set up");
    if (n != c->n_nodes) return fail(GFS_E_ARG, "order length mismatch");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    void *d_tmp = nullptr;
    HIPCHK(hipMalloc(&d_tmp, n * (2 * 8 + 2 * 4)));
    uint32_t *d_order = nullptr;
    hipError_t e = gfs::sort_order_device(c->d_x, c->d_perm, n, 1, d_tmp, &d_order);
    if (e != hipSuccess) { (void)hipFree(d_tmp); return fail(GFS_E_HIP, std::string("sort_order_device: ") + hipGetErrorString(e)); }
    std::vector<uint32_t> tmp(n);
    e = hipMemcpy(tmp.data(), d_order, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail(GFS_E_HIP, std::string("hipMemcpy order: ") + hipGetErrorString(e));
    for (uint64_t k = 0; k < n; ++k) order[k] = tmp[k];
    return GFS_OK;
}

// ---- multi-GPU replica merge helpers (device pointers, asynchronous on hip_stream) -----------------
int gfs_merge_prepare(const double *x, const double *x_prev, float *buf2n, uint64_t n, void *hip_stream) {
    if (!x || !x_prev || !buf2n) return fail(GFS_E_ARG, "null argument");
    if (n == 0) return GFS_OK;
    hipError_t e = gfs::launch_merge_prepare(x, x_prev, buf2n, n, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(GFS_E_HIP, std::string("merge_prepare: ") + hipGetErrorString(e));
    return GFS_OK;
}
int gfs_merge_apply(double *x, double *x_prev, const float *buf2n, uint64_t n, double divide_all_by, void *hip_stream) {
    if (!x || !x_prev || !buf2n) return fail(GFS_E_ARG, "null argument");
    if (n == 0) return GFS_OK;
    hipError_t e = gfs::launch_merge_apply(x, x_prev, buf2n, n, divide_all_by, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(GFS_E_HIP, std::string("merge_apply: ") + hipGetErrorString(e));
    return GFS_OK;
}

// ---- one-shot entry points -------------------------------------------------------------------
static int one_shot(const gfs_graph_view *g, const gfs_sgd_params *p, int dims, const gfs_launch_config *cfg,
                    const double *etas, const double *zetas, int init_x, double *x, gfs_stats *stats,
                    uint64_t *order = nullptr) {
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (!g || !p) return fail(GFS_E_ARG, "null argument");
    if (g->n_nodes == 0) return GFS_NOTHING_TO_DO;                         // sgd.rs:242-244,780-782
    if (!x) return fail(GFS_E_ARG, "positions buffer is null");
    auto t0 = std::chrono::steady_clock::now();
    const bool timing = std::getenv("GFS_TIMING") != nullptr;        // phase times of the one-shot call on stderr
    auto lap = [&](const char *what) {
        if (timing) std::fprintf(stderr, "[gfasort_hip] %-10s %8.2f ms\n", what,
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    gfs_ctx *c = nullptr;
    int rc = gfs_ctx_create(g, 0, &c);
    if (rc) return rc;
    lap("ctx_create");
    if (dims == 0) rc = gfs_ctx_setup_1d(c, p, cfg, etas, zetas);
    else { gfs_layout_params lp; lp.dimensions = (uint64_t)dims; lp.sgd = *p; rc = gfs_ctx_setup_nd(c, &lp, cfg, etas, zetas); }
    if (rc) { gfs_ctx_destroy(c); return rc; }
    lap("setup");
    if (dims == 0 && init_x) rc = gfs_ctx_init_positions(c);             // K4 on the device
    else rc = gfs_ctx_upload_positions(c, x, gfs_ctx_positions_len(c));
    lap("upload");
    if (!rc) rc = gfs_ctx_run(c, nullptr);
    lap("run");
    if (!rc) rc = gfs_ctx_download_positions(c, x, gfs_ctx_positions_len(c));
    lap("download");
    if (!rc && order) { rc = gfs_ctx_sort_order(c, order, g->n_nodes); lap("sort"); }
    if (!rc && stats) {
        rc = gfs_ctx_stats(c, stats);
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    gfs_ctx_destroy(c);
    lap("destroy");
    return rc;
}

int gfs_path_linear_sgd(const gfs_graph_view *g, const gfs_sgd_params *p, const gfs_launch_config *cfg,
                        const double *etas, const double *zetas, int init_x, double *x_inout, gfs_stats *stats) {
    return one_shot(g, p, 0, cfg, etas, zetas, init_x, x_inout, stats);
}

int gfs_path_sgd_sort(const gfs_graph_view *g, const gfs_sgd_params *p, const gfs_launch_config *cfg,
                      const double *etas, const double *zetas, int init_x, double *x_inout, uint64_t *order_out,
                      gfs_stats *stats) {
    if (!order_out) return fail(GFS_E_ARG, "order buffer is null");
    return one_shot(g, p, 0, cfg, etas, zetas, init_x, x_inout, stats, order_out);
}

int gfs_path_linear_sgd_layout(const gfs_graph_view *g, const gfs_layout_params *p, const gfs_launch_config *cfg,
                               const double *etas, const double *zetas, double *coords_inout, gfs_stats *stats) {
    if (!p) return fail(GFS_E_ARG, "params is null");
    if (p->dimensions < 1 || p->dimensions > GFS_MAX_DIMS) return fail(GFS_E_UNSUPPORTED, "dimensions must be 1..8");
    return one_shot(g, &p->sgd, (int)p->dimensions, cfg, etas, zetas, 0, coords_inout, stats);
}

}  // extern "C"
