// multi.hip — multi-device runs below the C ABI (include/gfasort_hip.h, "multi-device" section).
//
// The reference is one process with Hogwild threads on one shared vector (src/sgd.rs:413-593); this is its
// scale-out over the GPUs of a node: paths are sharded over ranks (one rank = one gfs_rank on
// one device), every rank runs its share of an iteration's term updates on its own replica of the positions,
// and the replicas are merged after every window of iterations.
//
// What is exchanged.  A rank reads and moves only the nodes its own paths step on.  In the node layout all
// ranks share (first-visit path order of the WHOLE graph) those nodes lie in a span of slots; a slot inside
// one rank's span only is moved by that rank alone and read by that rank alone, so it needs no exchange
// until the very end.  Only the slots covered by two or more spans are merged per window: the library
// compacts them into one buffer [delta | touched] that the caller all-reduces (sum) — RCCL over xGMI through
// torch.distributed, or ncclAllReduce called from the Rust host.  With paths that follow the graph (window
// graphs, chromosomes, contigs) that is a small fraction of the vector (C5 at 8 ranks: 7 %); when every path
// spans the whole graph it is the whole vector, as in round 1.  At the end one full-length f64 all-reduce of
// "what I own" completes every replica (gfs_rank_finish_*).
//
// Host-only planning (gfs_shard_paths, gfs_shard_quotas, gfs_shared_node_layout, gfs_exchange_plan) needs no
// device and is what the CPU tests and the Python driver for test engines use too.
#include "../../include/gfasort_hip.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

int gfs_set_error(int code, const std::string &msg);      // capi.hip

#define MHIPCHK(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return gfs_set_error(GFS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

namespace gfs {

struct ESeg { uint64_t lo, hi, off; };      // elements [lo, hi) of the position vector <-> buffer entries [off, off + hi - lo)

__device__ __forceinline__ uint32_t find_seg(const ESeg *segs, uint32_t n, uint64_t e) {
    // last segment with off <= e (offsets ascend)
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (segs[mid].off <= e) lo = mid; else hi = mid; }
    return lo;
}

// x_prev[e] = x[elem(e)]
__global__ void exchange_snapshot_kernel(const double *x, double *x_prev, const ESeg *segs, uint32_t n_segs, uint64_t total) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const ESeg s = segs[find_seg(segs, n_segs, e)];
        x_prev[e] = x[s.lo + (e - s.off)];
    }
}
// buf[e] = x - x_prev (this rank's moves since the last merge); buf[total + e] = moved at all
template <typename T>
__global__ void exchange_prepare_kernel(const double *x, const double *x_prev, T *buf, const ESeg *segs, uint32_t n_segs, uint64_t total) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const ESeg s = segs[find_seg(segs, n_segs, e)];
        const double d = x[s.lo + (e - s.off)] - x_prev[e];
        buf[e] = (T)d;
        buf[total + e] = d != 0.0 ? (T)1 : (T)0;
    }
}
// x_prev += sum_delta / divisor; x = x_prev.  rule 3: divisor = max(1, ranks that moved the node); 1: 1; 2: world;
// 0: max(1, ranks that moved the node x cscale), cscale in (0, 1] falling with the window's learning rate (gfs_rank_window_end)
template <typename T>
__global__ void exchange_apply_kernel(double *x, double *x_prev, const T *buf, const ESeg *segs, uint32_t n_segs, uint64_t total,
                                      double divide_all_by, double cscale) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const ESeg s = segs[find_seg(segs, n_segs, e)];
        const double c = (double)buf[total + e] * cscale;
        const double div = divide_all_by > 0.0 ? divide_all_by : (c > 1.0 ? c : 1.0);
        const double v = x_prev[e] + (double)buf[e] / div;
        x_prev[e] = v;
        x[s.lo + (e - s.off)] = v;
    }
}
// y[k] = x[k] where this rank is the designated owner of element k, else 0 (owned: sorted disjoint [lo, hi) intervals)
__global__ void finish_mask_kernel(const double *x, double *y, const ESeg *owned, uint32_t n_owned, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        uint32_t lo = 0, hi = n_owned;
        bool mine = false;
        if (n_owned) {
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (owned[mid].lo <= k) lo = mid; else hi = mid; }
            mine = owned[lo].lo <= k && k < owned[lo].hi;
        }
        y[k] = mine ? x[k] : 0.0;
    }
}

}  // namespace gfs

// ---------------------------------------------------------------------------------------------
// host-only planning
// ---------------------------------------------------------------------------------------------
extern "C" {

// Which rank owns which path.  Weight of a path = its step count if it has more than one step, else 0 (a one-step
// path never yields a term, sgd.rs:448).  sharding 1: consecutive blocks of paths with nearly equal weight (a rank's
// paths, and so the nodes it moves, stay together when the input's paths follow the graph); 2: longest-first bin
// packing; 0: blocks when they are balanced to 10 %, else bin packing.  rank_steps[r] = weight owned by r.
int gfs_shard_paths(const gfs_graph_view *g, uint32_t world, uint32_t sharding, uint32_t *path_owner, uint64_t *rank_steps) {
    if (!g || !rank_steps || (!path_owner && g->n_paths) || !g->path_first_step) return gfs_set_error(GFS_E_ARG, "null argument");
    if (world == 0 || sharding > 2) return gfs_set_error(GFS_E_ARG, "world must be >= 1, sharding 0..2");
    const uint64_t P = g->n_paths;
    std::vector<uint64_t> w(P);
    uint64_t total = 0;
    for (uint64_t p = 0; p < P; ++p) {
        const uint64_t c = g->path_first_step[p + 1] - g->path_first_step[p];
        w[p] = c > 1 ? c : 0; total += w[p];
    }
    std::fill(rank_steps, rank_steps + world, (uint64_t)0);
    auto contiguous = [&]() {
        // the path whose cumulative-weight midpoint falls into the r-th 1/world of the total goes to rank r
        uint64_t cum = 0; uint32_t last = 0;
        for (uint64_t p = 0; p < P; ++p) {
            uint32_t r = last;
            if (total && w[p]) {
                const long double mid = (long double)cum + (long double)w[p] / 2.0L;
                r = (uint32_t)std::min<long double>((long double)(world - 1), mid * world / (long double)total);
            }
            path_owner[p] = r; last = r; cum += w[p];
        }
    };
    auto loads = [&]() { std::fill(rank_steps, rank_steps + world, (uint64_t)0); for (uint64_t p = 0; p < P; ++p) rank_steps[path_owner[p]] += w[p]; };
    bool use_lpt = sharding == 2;
    if (!use_lpt) {
        contiguous(); loads();
        if (sharding == 0 && total) {
            const uint64_t mx = *std::max_element(rank_steps, rank_steps + world);
            if ((long double)mx * world > 1.10L * (long double)total) use_lpt = true;
        }
    }
    if (use_lpt) {
        std::vector<uint64_t> order(P);
        std::iota(order.begin(), order.end(), (uint64_t)0);
        std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return w[a] > w[b]; });
        std::vector<uint64_t> load(world, 0);
        for (uint64_t p : order) {
            const uint32_t r = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
            path_owner[p] = r; load[r] += w[p];
        }
        loads();
    }
    return GFS_OK;
}

// An iteration's term updates split in proportion to the ranks' step weights (largest remainder): sums exactly.
int gfs_shard_quotas(uint64_t updates, const uint64_t *rank_steps, uint32_t world, uint64_t *rank_quota) {
    if (!rank_steps || !rank_quota || world == 0) return gfs_set_error(GFS_E_ARG, "null argument");
    unsigned __int128 S = 0;
    for (uint32_t r = 0; r < world; ++r) S += rank_steps[r];
    if (S == 0) { std::fill(rank_quota, rank_quota + world, (uint64_t)0); return GFS_OK; }
    uint64_t given = 0;
    std::vector<std::pair<unsigned __int128, uint32_t>> rem(world);
    for (uint32_t r = 0; r < world; ++r) {
        const unsigned __int128 num = (unsigned __int128)updates * rank_steps[r];
        rank_quota[r] = (uint64_t)(num / S); given += rank_quota[r];
        rem[r] = { num % S, r };
    }
    std::stable_sort(rem.begin(), rem.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
    for (uint64_t k = 0; k < updates - given; ++k) rank_quota[rem[k].second]++;
    return GFS_OK;
}

// perm[k] = slot of dense node k — the rule gfs_ctx_create applies to ITS graph (index_kernels.hip, which explains it), here
// on the whole graph and on the host, so that all ranks store their replicas alike: nodes in the order the paths first step
// on them, except that a run of first visits which does not start its path is placed right after the root-run node it
// branches off (through its chain of anchors); unvisited nodes last, in index order.
int gfs_shared_node_layout(const gfs_graph_view *g, uint32_t *perm) {
    if (!g || (!perm && g->n_nodes)) return gfs_set_error(GFS_E_ARG, "null argument");
    const uint64_t N = g->n_nodes, S = g->n_steps;
    constexpr uint64_t NONE = ~0ull;
    std::vector<uint64_t> first(N, NONE);
    for (uint64_t s = 0; s < S; ++s) {
        const uint32_t n = g->step_node[s];
        if (n == GFS_NO_NODE) continue;
        if (n >= N) return gfs_set_error(GFS_E_ARG, "step_node out of range");
        if (first[n] == NONE) first[n] = s;
    }
    // one pass in step order: the run a first visit belongs to, its anchor, and — anchors are always visited earlier — its root
    std::vector<uint32_t> root(N);
    for (uint64_t k = 0; k < N; ++k) root[k] = (uint32_t)k;
    uint64_t p = 0;                                                    // path of step s
    bool prev_first = false;                                           // step s-1 was a first visit (same path)
    uint32_t run_anchor = GFS_NO_NODE;                                 // NO_NODE: the current run is a root run
    for (uint64_t s = 0; s < S; ++s) {
        while (p + 1 < g->n_paths && g->path_first_step[p + 1] <= s) ++p;
        const bool start = g->n_paths && g->path_first_step[p] == s;
        if (start) prev_first = false;
        const uint32_t n = g->step_node[s];
        const bool is_first = n != GFS_NO_NODE && first[n] == s;
        if (is_first) {
            if (!prev_first) run_anchor = start ? GFS_NO_NODE : g->step_node[s - 1];          // a new run (its anchor may be absent)
            if (run_anchor != GFS_NO_NODE) root[n] = root[run_anchor];
        }
        prev_first = is_first;
    }
    struct Key { uint64_t key, first; uint32_t id; };
    std::vector<Key> keys(N);
    for (uint64_t k = 0; k < N; ++k) {
        const uint64_t f = first[k];
        keys[k] = {f == NONE ? NONE : (root[k] == k ? 2 * f : 2 * first[root[k]] + 1), f, (uint32_t)k};
    }
    std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
        if (a.key != b.key) return a.key < b.key;
        if (a.first != b.first) return a.first < b.first;
        return a.id < b.id;
    });
    for (uint64_t r = 0; r < N; ++r) perm[keys[r].id] = (uint32_t)r;
    return GFS_OK;
}

// Spans and what must be exchanged.  span_lo/hi[r]: the slots [lo, hi) rank r's multi-step paths touch (lo == hi: none).
// seg_lo/hi[0..*n_seg): the maximal slot intervals covered by two or more spans, ascending (at most world - 1 of them...
// at most 2*world entries are ever written).  own_lo/hi/own_rank[0..*n_own): maximal intervals with their designated
// owner = the lowest rank whose span covers them; slots no span covers (they never move) belong to rank 0, so that the
// intervals tile [0, n_nodes): at most 2*world + 1 entries, the arrays must hold 2*world + 2.
int gfs_exchange_plan(const gfs_graph_view *g, const uint32_t *perm, const uint32_t *path_owner, uint32_t world,
                      uint64_t *span_lo, uint64_t *span_hi, uint64_t *seg_lo, uint64_t *seg_hi, uint32_t *n_seg,
                      uint64_t *own_lo, uint64_t *own_hi, uint32_t *own_rank, uint32_t *n_own) {
    if (!g || !perm || (!path_owner && g->n_paths) || !span_lo || !span_hi || !seg_lo || !seg_hi || !n_seg)
        return gfs_set_error(GFS_E_ARG, "null argument");
    if (world == 0) return gfs_set_error(GFS_E_ARG, "world must be >= 1");
    std::vector<uint64_t> lo(world, UINT64_MAX), hi(world, 0);
    for (uint64_t p = 0; p < g->n_paths; ++p) {
        const uint64_t b = g->path_first_step[p], e = g->path_first_step[p + 1];
        if (e - b < 2) continue;
        const uint32_t r = path_owner[p];
        if (r >= world) return gfs_set_error(GFS_E_ARG, "path_owner out of range");
        uint64_t l = lo[r], h = hi[r];
        for (uint64_t s = b; s < e; ++s) {
            const uint32_t n = g->step_node[s];
            if (n == GFS_NO_NODE) continue;
            const uint64_t slot = perm[n];
            if (slot < l) l = slot;
            if (slot + 1 > h) h = slot + 1;
        }
        lo[r] = l; hi[r] = h;
    }
    struct Ev { uint64_t at; int d; uint32_t r; };
    std::vector<Ev> ev;
    for (uint32_t r = 0; r < world; ++r) {
        if (lo[r] == UINT64_MAX) { span_lo[r] = span_hi[r] = 0; continue; }
        span_lo[r] = lo[r]; span_hi[r] = hi[r];
        ev.push_back({ lo[r], +1, r }); ev.push_back({ hi[r], -1, r });
    }
    std::sort(ev.begin(), ev.end(), [](const Ev &a, const Ev &b) { return a.at != b.at ? a.at < b.at : a.d < b.d; });
    std::vector<uint8_t> active(world, 0);
    uint32_t cover = 0, ns = 0, no = 0;
    for (size_t i = 0; i < ev.size();) {
        const uint64_t at = ev[i].at;
        while (i < ev.size() && ev[i].at == at) { active[ev[i].r] = ev[i].d > 0; cover = (uint32_t)((int)cover + ev[i].d); ++i; }
        if (i == ev.size()) break;
        const uint64_t next = ev[i].at;
        if (next == at || cover == 0) continue;
        if (cover >= 2) {
            if (ns && seg_hi[ns - 1] == at) seg_hi[ns - 1] = next;
            else { seg_lo[ns] = at; seg_hi[ns] = next; ++ns; }
        }
        if (own_lo && own_hi && own_rank && n_own) {
            uint32_t owner = 0; while (!active[owner]) ++owner;
            if (no && own_hi[no - 1] == at && own_rank[no - 1] == owner) own_hi[no - 1] = next;
            else { own_lo[no] = at; own_hi[no] = next; own_rank[no] = owner; ++no; }
        }
    }
    *n_seg = ns;
    if (own_lo && own_hi && own_rank && n_own) {
        // Slots no span covers — nodes no path visits, nodes only one-step paths step on, gaps between spans — never move,
        // but gfs_rank_finish_* rebuilds every replica as the SUM of what the ranks own: a slot nobody owned came out as 0
        // where a world == 1 run (and the reference, sgd.rs:286-294) leaves the start position.  Rank 0 owns them, so that
        // the own_* intervals tile [0, n_nodes): at most 2*world + 1 of them.
        std::vector<uint64_t> lo2, hi2; std::vector<uint32_t> rk2;
        auto push = [&](uint64_t l, uint64_t h, uint32_t r) {
            if (l >= h) return;
            if (!lo2.empty() && hi2.back() == l && rk2.back() == r) hi2.back() = h;
            else { lo2.push_back(l); hi2.push_back(h); rk2.push_back(r); }
        };
        uint64_t at = 0;
        for (uint32_t k = 0; k < no; ++k) { push(at, own_lo[k], 0u); push(own_lo[k], own_hi[k], own_rank[k]); at = own_hi[k]; }
        push(at, g->n_nodes, 0u);
        no = (uint32_t)lo2.size();
        for (uint32_t k = 0; k < no; ++k) { own_lo[k] = lo2[k]; own_hi[k] = hi2[k]; own_rank[k] = rk2[k]; }
        *n_own = no;
    }
    return GFS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// gfs_rank
// ---------------------------------------------------------------------------------------------
struct gfs_rank {
    gfs_rank_config cfg{};
    gfs_sgd_params params{};
    uint64_t dims = 0;
    gfs_ctx *ctx = nullptr;
    bool idle = false;                   // no term updates on this rank (no multi-step path, or a zero quota)
    uint64_t n_nodes = 0, x_len = 0;
    std::vector<uint64_t> quotas, rank_steps, span_lo, span_hi;
    uint64_t shared_slots = 0;           // slots more than one rank can move
    uint64_t total = 0;                  // exchanged elements (shared slots x planes x dims), 0 when world == 1
    std::vector<gfs::ESeg> esegs, owned;
    gfs::ESeg *d_esegs = nullptr, *d_owned = nullptr;
    double *d_xprev = nullptr;
    void *d_buf = nullptr; bool buf_owned = false;
    double *d_full = nullptr;            // finish: full-length scratch when the caller binds none
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double merge_ms = 0.0; uint64_t windows = 0; bool ev_pending = false;
    std::vector<double> etas;            // the schedule (merge rule 0)
    double eta_sum = 1.0;                // learning rate below which a window's moves are small enough to be summed (merge rule 0)
    uint64_t win_last_k = 0, win_len = 1;
};

static size_t payload_size(const gfs_rank *r) { return r->cfg.payload ? sizeof(double) : sizeof(float); }

extern "C" {

void gfs_rank_destroy(gfs_rank *r) {
    if (!r) return;
    (void)hipSetDevice(r->cfg.device);
    if (r->ctx) gfs_ctx_destroy(r->ctx);
    if (r->d_esegs) (void)hipFree(r->d_esegs);
    if (r->d_owned) (void)hipFree(r->d_owned);
    if (r->d_xprev) (void)hipFree(r->d_xprev);
    if (r->d_buf && r->buf_owned) (void)hipFree(r->d_buf);
    if (r->d_full) (void)hipFree(r->d_full);
    for (auto &e : r->ev) if (e) (void)hipEventDestroy(e);
    delete r;
}

int gfs_rank_create(const gfs_graph_view *g, const gfs_sgd_params *p, uint64_t dims, const gfs_rank_config *cfg, gfs_rank **out) {
    if (!g || !p || !cfg || !out) return gfs_set_error(GFS_E_ARG, "null argument");
    *out = nullptr;
    if (cfg->world == 0 || cfg->rank >= cfg->world) return gfs_set_error(GFS_E_ARG, "rank must be < world");
    if (cfg->world > GFS_MAX_WORLD) return gfs_set_error(GFS_E_UNSUPPORTED, "world too large");
    if (dims > GFS_MAX_DIMS) return gfs_set_error(GFS_E_UNSUPPORTED, "dimensions must be 1..8");
    if (cfg->merge_rule > 3 || cfg->payload > 1 || cfg->exchange > 1 || cfg->sharding > 2) return gfs_set_error(GFS_E_ARG, "bad rank config");
    gfs_rank *r = new (std::nothrow) gfs_rank();
    if (!r) return gfs_set_error(GFS_E_NOMEM, "out of memory");
    r->cfg = *cfg; r->params = *p; r->dims = dims; r->n_nodes = g->n_nodes;
    const uint32_t W = cfg->world;
    int rc = GFS_OK;
    auto bail = [&](int code) { gfs_rank_destroy(r); return code; };

    {   // merge rule 0: the schedule, and the scale of a short-range term (mean node length in bp)
        r->etas.assign(p->iter_max + 1, 0.0);
        gfs_sgd_schedule(p, r->etas.data());
        long double bp = 0; for (uint64_t k = 0; k < g->n_nodes; ++k) bp += g->node_len[k];
        r->eta_sum = g->n_nodes ? std::max<double>(1.0, (double)(bp / (long double)g->n_nodes)) : 1.0;
        if (const char *e = std::getenv("GFS_DBG_MERGE_ETA_FACTOR")) { const double f = std::atof(e); if (f > 0.0) r->eta_sum *= f; }   // probe knob
    }
    // plan: owners, quotas, shared layout, spans
    std::vector<uint32_t> owner(std::max<uint64_t>(g->n_paths, 1));
    r->rank_steps.assign(W, 0); r->quotas.assign(W, 0); r->span_lo.assign(W, 0); r->span_hi.assign(W, 0);
    if ((rc = gfs_shard_paths(g, W, cfg->sharding, owner.data(), r->rank_steps.data()))) return bail(rc);
    if ((rc = gfs_shard_quotas(p->min_term_updates, r->rank_steps.data(), W, r->quotas.data()))) return bail(rc);
    std::vector<uint32_t> perm(std::max<uint64_t>(g->n_nodes, 1));
    const bool multi = W > 1;
    std::vector<uint64_t> seg_lo(2 * W), seg_hi(2 * W), own_lo(2 * W + 2), own_hi(2 * W + 2);
    std::vector<uint32_t> own_rank(2 * W + 2);
    uint32_t n_seg = 0, n_own = 0;
    if (multi) {
        if ((rc = gfs_shared_node_layout(g, perm.data()))) return bail(rc);
        if ((rc = gfs_exchange_plan(g, perm.data(), owner.data(), W, r->span_lo.data(), r->span_hi.data(), seg_lo.data(), seg_hi.data(),
                                    &n_seg, own_lo.data(), own_hi.data(), own_rank.data(), &n_own))) return bail(rc);
        if (cfg->exchange == 1 && g->n_nodes) { n_seg = 1; seg_lo[0] = 0; seg_hi[0] = g->n_nodes; }
    }

    // this rank's shard: its paths' steps, all nodes
    std::vector<uint32_t> step_node; std::vector<uint8_t> step_rev; std::vector<uint64_t> first(1, 0);
    gfs_graph_view local = *g;
    if (multi) {
        uint64_t S = 0;
        for (uint64_t q = 0; q < g->n_paths; ++q) if (owner[q] == cfg->rank) S += g->path_first_step[q + 1] - g->path_first_step[q];
        step_node.reserve(S); step_rev.reserve(S);
        for (uint64_t q = 0; q < g->n_paths; ++q) {
            if (owner[q] != cfg->rank) continue;
            const uint64_t b = g->path_first_step[q], e = g->path_first_step[q + 1];
            step_node.insert(step_node.end(), g->step_node + b, g->step_node + e);
            step_rev.insert(step_rev.end(), g->step_is_rev + b, g->step_is_rev + e);
            first.push_back(step_node.size());
        }
        local.n_steps = step_node.size(); local.n_paths = first.size() - 1;
        static const uint32_t z32 = 0; static const uint8_t z8 = 0;
        local.step_node = step_node.empty() ? &z32 : step_node.data();
        local.step_is_rev = step_rev.empty() ? &z8 : step_rev.data();
        local.path_first_step = first.data();
    }
    if ((rc = gfs_ctx_create_with_layout(&local, cfg->device, multi ? perm.data() : nullptr, &r->ctx))) return bail(rc);
    gfs_launch_config lc = cfg->launch;
    lc.term_updates_per_iteration = multi ? r->quotas[cfg->rank] : cfg->launch.term_updates_per_iteration;
    lc.stream_base = (uint64_t)cfg->rank * (lc.n_streams ? lc.n_streams : (1ull << 20)) + cfg->launch.stream_base;
    r->idle = multi && r->quotas[cfg->rank] == 0;
    if (r->idle) lc.term_updates_per_iteration = 1;                       // placeholder: an idle rank never launches
    if (dims == 0) rc = gfs_ctx_setup_1d(r->ctx, p, &lc, nullptr, nullptr);
    else { gfs_layout_params lp; lp.dimensions = dims; lp.sgd = *p; rc = gfs_ctx_setup_nd(r->ctx, &lp, &lc, nullptr, nullptr); }
    if (rc < 0) return bail(rc);
    if (rc == GFS_NOTHING_TO_DO) r->idle = true;
    r->x_len = gfs_ctx_positions_len(r->ctx);

    // exchange tables (element space: 1D x[slot]; nD the end x dimension planes coords[end][dim][slot], sgd_device.h coord_ptr)
    if (multi && r->x_len) {
        const uint64_t planes = dims ? 2ull * dims : 1ull, N = g->n_nodes;
        uint64_t off = 0;
        for (uint64_t pl = 0; pl < planes; ++pl)
            for (uint32_t k = 0; k < n_seg; ++k) {
                r->esegs.push_back({ pl * N + seg_lo[k], pl * N + seg_hi[k], off });
                off += seg_hi[k] - seg_lo[k];
            }
        r->total = off;
        for (uint32_t k = 0; k < n_seg; ++k) r->shared_slots += seg_hi[k] - seg_lo[k];
        for (uint64_t pl = 0; pl < planes; ++pl)
            for (uint32_t k = 0; k < n_own; ++k)
                if (own_rank[k] == cfg->rank) r->owned.push_back({ pl * N + own_lo[k], pl * N + own_hi[k], 0 });
        if (hipSetDevice(cfg->device) != hipSuccess) return bail(gfs_set_error(GFS_E_HIP, "hipSetDevice failed"));
        auto up = [&](const std::vector<gfs::ESeg> &v, gfs::ESeg **d) -> int {
            if (v.empty()) return GFS_OK;
            MHIPCHK(hipMalloc(d, v.size() * sizeof(gfs::ESeg)));
            MHIPCHK(hipMemcpy(*d, v.data(), v.size() * sizeof(gfs::ESeg), hipMemcpyHostToDevice));
            return GFS_OK;
        };
        if ((rc = up(r->esegs, &r->d_esegs)) || (rc = up(r->owned, &r->d_owned))) return bail(rc);
        if (r->total) {
            if (hipMalloc(&r->d_xprev, r->total * 8) != hipSuccess) return bail(gfs_set_error(GFS_E_NOMEM, "hipMalloc x_prev"));
            if (hipMemset(r->d_xprev, 0, r->total * 8) != hipSuccess) return bail(gfs_set_error(GFS_E_HIP, "hipMemset x_prev"));
        }
        for (auto &e : r->ev) if (hipEventCreate(&e) != hipSuccess) return bail(gfs_set_error(GFS_E_HIP, "hipEventCreate"));
    }
    *out = r;
    return r->idle && !multi ? GFS_NOTHING_TO_DO : GFS_OK;
}

gfs_ctx *gfs_rank_ctx(gfs_rank *r) { return r ? r->ctx : nullptr; }

int gfs_rank_get_info(const gfs_rank *r, gfs_rank_info *out) {
    if (!r || !out) return gfs_set_error(GFS_E_ARG, "null argument");
    std::memset(out, 0, sizeof *out);
    out->quota = r->cfg.world > 1 ? r->quotas[r->cfg.rank] : r->params.min_term_updates;
    out->shard_steps = r->rank_steps.empty() ? 0 : r->rank_steps[r->cfg.rank];
    out->span_lo = r->span_lo.empty() ? 0 : r->span_lo[r->cfg.rank];
    out->span_hi = r->span_hi.empty() ? 0 : r->span_hi[r->cfg.rank];
    out->shared_slots = r->shared_slots;
    out->exchange_count = 2 * r->total;
    out->positions_len = r->x_len;
    out->idle = r->idle ? 1 : 0;
    out->windows = r->windows;
    if (r->ev_pending) {
        // the last window's merge kernels (events recorded on the caller's stream)
        float a = 0.f, b = 0.f;
        if (hipEventSynchronize(r->ev[3]) == hipSuccess && hipEventElapsedTime(&a, r->ev[0], r->ev[1]) == hipSuccess &&
            hipEventElapsedTime(&b, r->ev[2], r->ev[3]) == hipSuccess) out->last_merge_kernels_ms = a + b;
    }
    return GFS_OK;
}

uint64_t gfs_rank_exchange_count(const gfs_rank *r) { return r ? 2 * r->total : 0; }

static int ensure_buf(gfs_rank *r) {
    if (r->d_buf || r->total == 0) return GFS_OK;
    MHIPCHK(hipSetDevice(r->cfg.device));
    MHIPCHK(hipMalloc(&r->d_buf, 2 * r->total * payload_size(r)));
    r->buf_owned = true;
    return GFS_OK;
}
void *gfs_rank_exchange_buffer(gfs_rank *r) {
    if (!r || ensure_buf(r)) return nullptr;
    return r->d_buf;
}
int gfs_rank_bind_exchange_buffer(gfs_rank *r, void *device_ptr) {
    if (!r || !device_ptr) return gfs_set_error(GFS_E_ARG, "null argument");
    if (r->d_buf && r->buf_owned) { MHIPCHK(hipSetDevice(r->cfg.device)); MHIPCHK(hipFree(r->d_buf)); }
    r->d_buf = device_ptr; r->buf_owned = false;
    return GFS_OK;
}

static int snapshot(gfs_rank *r, hipStream_t st) {
    if (!r->total) return GFS_OK;
    MHIPCHK(hipSetDevice(r->cfg.device));
    hipLaunchKernelGGL(gfs::exchange_snapshot_kernel, dim3(1024), dim3(256), 0, st, (const double *)gfs_ctx_positions_device(r->ctx),
                       r->d_xprev, r->d_esegs, (uint32_t)r->esegs.size(), r->total);
    MHIPCHK(hipGetLastError());
    return GFS_OK;
}

// host == NULL: the reference's start (1D only)
int gfs_rank_set_positions(gfs_rank *r, const double *host, uint64_t n) {
    if (!r) return gfs_set_error(GFS_E_ARG, "rank is null");
    int rc = host ? gfs_ctx_upload_positions(r->ctx, host, n) : gfs_ctx_init_positions(r->ctx);
    if (rc) return rc;
    rc = snapshot(r, nullptr);
    if (rc) return rc;
    MHIPCHK(hipDeviceSynchronize());
    return GFS_OK;
}
// the caller replaced the positions behind the library's back (gfs_ctx_bind_positions + its own copy): re-snapshot
int gfs_rank_positions_changed(gfs_rank *r, void *hip_stream) {
    if (!r) return gfs_set_error(GFS_E_ARG, "rank is null");
    return snapshot(r, (hipStream_t)hip_stream);
}
int gfs_rank_get_positions(gfs_rank *r, double *host, uint64_t n) {
    if (!r) return gfs_set_error(GFS_E_ARG, "rank is null");
    return gfs_ctx_download_positions(r->ctx, host, n);
}

// One merge window, first half: this rank's share of iterations ks[0..n) (one fused launch where possible), then its
// moves of the shared slots into the exchange buffer.  The caller all-reduces (sum) gfs_rank_exchange_count() elements
// of the payload type in gfs_rank_exchange_buffer() on the same stream, then calls gfs_rank_window_end.
int gfs_rank_window_begin(gfs_rank *r, const uint64_t *ks, uint64_t n, void *hip_stream) {
    if (!r || (!ks && n)) return gfs_set_error(GFS_E_ARG, "null argument");
    hipStream_t st = (hipStream_t)hip_stream;
    if (n) { r->win_last_k = ks[n - 1]; r->win_len = n; }
    if (!r->idle && n) {
        const int rc = gfs_ctx_run_range(r->ctx, ks, n, hip_stream);
        if (rc < 0) return rc;
    }
    if (!r->total) return GFS_OK;
    int rc = ensure_buf(r);
    if (rc) return rc;
    MHIPCHK(hipSetDevice(r->cfg.device));
    const double *x = (const double *)gfs_ctx_positions_device(r->ctx);
    MHIPCHK(hipEventRecord(r->ev[0], st));
    if (r->cfg.payload) hipLaunchKernelGGL((gfs::exchange_prepare_kernel<double>), dim3(1024), dim3(256), 0, st, x, r->d_xprev,
                                           (double *)r->d_buf, r->d_esegs, (uint32_t)r->esegs.size(), r->total);
    else hipLaunchKernelGGL((gfs::exchange_prepare_kernel<float>), dim3(1024), dim3(256), 0, st, x, r->d_xprev, (float *)r->d_buf,
                            r->d_esegs, (uint32_t)r->esegs.size(), r->total);
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipEventRecord(r->ev[1], st));
    return GFS_OK;
}
int gfs_rank_window_end(gfs_rank *r, void *hip_stream) {
    if (!r) return gfs_set_error(GFS_E_ARG, "rank is null");
    r->windows++;
    if (!r->total) return GFS_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    MHIPCHK(hipSetDevice(r->cfg.device));
    double *x = (double *)gfs_ctx_positions_device(r->ctx);
    const double div = (r->cfg.merge_rule == 0 || r->cfg.merge_rule == 3) ? 0.0 : (r->cfg.merge_rule == 1 ? 1.0 : (double)r->cfg.world);   // 0 annealed, 3 plain touch
    // Rule 0 (annealed, the default).  c ranks each ran their share of the window on their own replica.  While the learning rate is large a
    // term is a FULL correction of its pair (mu = min(eta / d, 1) = 1): c replicas propose c full corrections of the same error
    // and only their mean is safe (rule 3, "touch"; their sum diverges).  Once eta is below the distance of even the shortest terms every
    // move is a small step (mu << 1) and the steps of all ranks ADD, as they do on one shared vector — averaging them throws
    // away (c - 1)/c of the window's work exactly where the layout is being finished: 8 ranks under rule 3 left the 525k-node
    // bubble graph at 1.47x the single-GPU error at path distance 1 (profiles/r03/virtual_cluster_touch_rule.log).  The divisor
    // therefore falls from c to 1 with eta: max(1, c * min(1, window length * eta / mean node length)).
    double cscale = 1.0;
    if (r->cfg.merge_rule == 0 && r->win_last_k < r->etas.size())
        cscale = std::min(1.0, (double)r->win_len * r->etas[r->win_last_k] / r->eta_sum);
    MHIPCHK(hipEventRecord(r->ev[2], st));
    if (r->cfg.payload) hipLaunchKernelGGL((gfs::exchange_apply_kernel<double>), dim3(1024), dim3(256), 0, st, x, r->d_xprev,
                                           (const double *)r->d_buf, r->d_esegs, (uint32_t)r->esegs.size(), r->total, div, cscale);
    else hipLaunchKernelGGL((gfs::exchange_apply_kernel<float>), dim3(1024), dim3(256), 0, st, x, r->d_xprev, (const float *)r->d_buf,
                            r->d_esegs, (uint32_t)r->esegs.size(), r->total, div, cscale);
    MHIPCHK(hipGetLastError());
    MHIPCHK(hipEventRecord(r->ev[3], st));
    r->ev_pending = true;
    return GFS_OK;
}

// End of the run: a slot inside one span only is current on that rank alone.  begin: full[k] = x[k] where this rank is
// the slot's designated owner, else 0; the caller all-reduces (sum, f64) the positions_len elements; end: x = full.
int gfs_rank_finish_begin(gfs_rank *r, double *full_device, void *hip_stream) {
    if (!r) return gfs_set_error(GFS_E_ARG, "rank is null");
    if (r->cfg.world < 2 || !r->x_len) return GFS_OK;
    MHIPCHK(hipSetDevice(r->cfg.device));
    if (!full_device) {
        if (!r->d_full) MHIPCHK(hipMalloc(&r->d_full, r->x_len * 8));
        full_device = r->d_full;
    }
    hipLaunchKernelGGL(gfs::finish_mask_kernel, dim3(2048), dim3(256), 0, (hipStream_t)hip_stream,
                       (const double *)gfs_ctx_positions_device(r->ctx), full_device, r->d_owned, (uint32_t)r->owned.size(), r->x_len);
    MHIPCHK(hipGetLastError());
    return GFS_OK;
}
double *gfs_rank_finish_buffer(gfs_rank *r) { return r ? r->d_full : nullptr; }
int gfs_rank_finish_end(gfs_rank *r, const double *full_device, void *hip_stream) {
    if (!r) return gfs_set_error(GFS_E_ARG, "rank is null");
    if (r->cfg.world < 2 || !r->x_len) return GFS_OK;
    if (!full_device) full_device = r->d_full;
    if (!full_device) return gfs_set_error(GFS_E_STATE, "gfs_rank_finish_begin has not run");
    MHIPCHK(hipSetDevice(r->cfg.device));
    MHIPCHK(hipMemcpyAsync(gfs_ctx_positions_device(r->ctx), full_device, r->x_len * 8, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    return snapshot(r, (hipStream_t)hip_stream);
}

// The whole schedule with a caller-supplied collective: windows of merge_every iterations (the last iteration always
// merges), then the final completion.  allreduce(user, device_buf, count, is_f64, hip_stream) must sum the buffer over
// all ranks in place, ordered on hip_stream.  world == 1: no collective is called.
int gfs_rank_run(gfs_rank *r, gfs_allreduce_fn allreduce, void *user, void *hip_stream) {
    if (!r) return gfs_set_error(GFS_E_ARG, "rank is null");
    if (r->cfg.world > 1 && !allreduce) return gfs_set_error(GFS_E_ARG, "a collective is needed for world > 1");
    const uint64_t iters = r->params.iter_max + 1, every = r->cfg.merge_every ? r->cfg.merge_every : 1;
    std::vector<uint64_t> ks;
    for (uint64_t k0 = 0; k0 < iters; k0 += every) {
        ks.clear();
        for (uint64_t k = k0; k < std::min(iters, k0 + every); ++k) ks.push_back(k);
        int rc = gfs_rank_window_begin(r, ks.data(), ks.size(), hip_stream);
        if (rc < 0) return rc;
        if (r->cfg.world > 1) {
            if (r->total && (rc = allreduce(user, r->d_buf, 2 * r->total, (int)r->cfg.payload, hip_stream)) != 0)
                return gfs_set_error(GFS_E_STATE, "the caller's all-reduce failed");
            if ((rc = gfs_rank_window_end(r, hip_stream)) < 0) return rc;
        }
    }
    if (r->cfg.world > 1) {
        int rc = gfs_rank_finish_begin(r, nullptr, hip_stream);
        if (rc < 0) return rc;
        if (r->x_len && (rc = allreduce(user, r->d_full, r->x_len, 1, hip_stream)) != 0)
            return gfs_set_error(GFS_E_STATE, "the caller's all-reduce failed");
        if ((rc = gfs_rank_finish_end(r, nullptr, hip_stream)) < 0) return rc;
    }
    MHIPCHK(hipSetDevice(r->cfg.device));
    MHIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
    return GFS_OK;
}

}  // extern "C"
