// sgd.cpp — see sgd.hpp.
#include "sgd.hpp"

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <iostream>
#include <ostream>
#include <stdexcept>

namespace gfasort {

gfs_sgd_params PathSGDParams::to_c() const {
    gfs_sgd_params c;
    std::memset(&c, 0, sizeof c);
    c.iter_max = iter_max; c.iter_with_max_learning_rate = iter_with_max_learning_rate;
    c.min_term_updates = min_term_updates; c.delta = delta; c.eps = eps; c.eta_max = eta_max; c.theta = theta;
    c.space = space; c.space_max = space_max; c.space_quantization_step = space_quantization_step;
    c.cooling_start = cooling_start; c.nthreads = nthreads; c.progress = progress ? 1 : 0; c.seed = seed;
    return c;
}

gfs_layout_params LayoutSGDParams::to_c() const {
    PathSGDParams s;
    s.iter_max = iter_max; s.iter_with_max_learning_rate = iter_with_max_learning_rate;
    s.min_term_updates = min_term_updates; s.delta = delta; s.eps = eps; s.eta_max = eta_max; s.theta = theta;
    s.space = space; s.space_max = space_max; s.space_quantization_step = space_quantization_step;
    s.cooling_start = cooling_start; s.nthreads = nthreads; s.progress = progress; s.seed = seed;
    gfs_layout_params c;
    c.dimensions = dimensions; c.sgd = s.to_c();
    return c;
}

// path statistics used by both from_graph functions (they each build a throw-away PathIndex
// in the reference: ygs.rs:58, sgd.rs:734)
static void path_stats(const BidirectedGraph &g, uint64_t &sum_steps, size_t &max_steps, size_t &max_len) {
    sum_steps = 0; max_steps = 0; max_len = 0;
    for (const auto &p : g.paths) {
        size_t pos = 0;
        for (Handle h : p.steps) {
            size_t id = h.node_id();
            if (id < g.nodes.size() && g.nodes[id].has_value()) pos += g.nodes[id]->sequence.size();
        }
        sum_steps += p.steps.size();
        max_steps = std::max(max_steps, p.steps.size());
        max_len = std::max(max_len, pos);
    }
}

LayoutSGDParams LayoutSGDParams::from_graph(const BidirectedGraph &g, size_t dimensions, size_t nthreads) {
    uint64_t sum; size_t mx, ml;
    path_stats(g, sum, mx, ml);
    LayoutSGDParams p;
    p.dimensions = dimensions; p.iter_max = 30; p.min_term_updates = 10 * sum;     // sgd.rs:749
    p.eta_max = (double)(mx * mx); p.space = mx; p.space_max = 1000; p.space_quantization_step = 100;
    p.nthreads = nthreads;
    return p;
}

YgsParams::YgsParams() {                                             // ygs.rs:23-45
    path_sgd.iter_max = 100; path_sgd.min_term_updates = 0; path_sgd.eta_max = 0.0; path_sgd.space = 0;
    path_sgd.space_max = 100; path_sgd.space_quantization_step = 100;
}

YgsParams YgsParams::from_graph(const BidirectedGraph &g, uint8_t verbose, size_t nthreads) {
    YgsParams y;
    y.verbose = verbose;
    y.path_sgd.nthreads = nthreads;
    y.path_sgd.progress = verbose >= 2;
    uint64_t sum; size_t mx, ml;
    path_stats(g, sum, mx, ml);
    y.path_sgd.min_term_updates = sum;                               // ygs.rs:73
    y.path_sgd.eta_max = (double)(mx * mx);                          // ygs.rs:76
    y.path_sgd.space = ml;                                           // ygs.rs:79
    if (verbose >= 2) {
        std::cerr << "[ygs_sort] Calculated parameters:\n  sum_path_step_count: " << sum
                  << "\n  max_path_step_count: " << mx << "\n  max_path_length: " << ml
                  << "\n  min_term_updates: " << y.path_sgd.min_term_updates
                  << "\n  eta_max: " << rust_display_f64(y.path_sgd.eta_max) << "\n  space: " << y.path_sgd.space << "\n";
    }
    return y;
}

// ---- Layout ---------------------------------------------------------------------------------------
Layout Layout::from_vectors(const std::vector<std::vector<double>> &v) {
    if (v.empty()) throw std::runtime_error("Must have at least 1 dimension");
    size_t entries = v[0].size();
    if (entries % 2) throw std::runtime_error("Must have even number of entries (2 per node)");
    for (const auto &d : v) if (d.size() != entries) throw std::runtime_error("All dimension vectors must have same length");
    Layout l(v.size(), entries / 2);
    for (size_t node = 0; node < l.num_nodes; ++node)
        for (size_t end = 0; end < 2; ++end)
            for (size_t dim = 0; dim < l.dimensions; ++dim) l.coords[l.index(node, end, dim)] = v[dim][node * 2 + end];
    return l;
}

double Layout::distance(size_t na, size_t ea, size_t nb, size_t eb) const {
    double s = 0.0;
    for (size_t d = 0; d < dimensions; ++d) { double delta = get(na, ea, d) - get(nb, eb, d); s += delta * delta; }
    return std::sqrt(s);
}

std::string rust_display_f64(double v) {
    if (v != v) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[512];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);     // shortest round-trip, no exponent
    return std::string(buf, r.ptr);
}

static const char *dim_name(size_t d) {                              // layout.rs:248-256
    switch (d) { case 0: return "x"; case 1: return "y"; case 2: return "z"; case 3: return "w"; default: return "d"; }
}

void Layout::write_tsv(std::ostream &out) const {
    out << "idx";
    for (size_t d = 0; d < dimensions; ++d) out << "\t" << dim_name(d) << "+";
    for (size_t d = 0; d < dimensions; ++d) out << "\t" << dim_name(d) << "-";
    out << "\n";
    // rows formatted in blocks by the host threads, a window of blocks at a time, written in order
    const size_t BLOCK = 1 << 14, nblocks = (num_nodes + BLOCK - 1) / BLOCK;
    const size_t window = std::max<size_t>(io_threads() * 4, 1);
    std::vector<std::string> piece(std::min(window, std::max<size_t>(nblocks, 1)));
    for (size_t base = 0; base < nblocks; base += window) {
        const size_t n = std::min(window, nblocks - base);
        parallel_for(n, [&](size_t i) {
            std::string &o = piece[i];
            o.clear();
            o.reserve(BLOCK * (8 + 2 * dimensions * 20));
            char buf[512];
            const size_t lo = (base + i) * BLOCK, hi = std::min(num_nodes, lo + BLOCK);
            for (size_t node = lo; node < hi; ++node) {
                auto r = std::to_chars(buf, buf + sizeof buf, node);
                o.append(buf, r.ptr);
                for (size_t end = 0; end < 2; ++end)
                    for (size_t d = 0; d < dimensions; ++d) {
                        o.push_back('\t');
                        const double v = get(node, end, d);
                        if (v != v || std::isinf(v)) { o += rust_display_f64(v); continue; }
                        auto rv = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
                        o.append(buf, rv.ptr);
                    }
                o.push_back('\n');
            }
        });
        for (size_t i = 0; i < n; ++i) out.write(piece[i].data(), (std::streamsize)piece[i].size());
    }
}

// ---- SGD entry points ------------------------------------------------------------------------------
static void check(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("gfasort_hip: ") + gfs_last_error());
}

// positions by dense index; empty when the reference returns an empty map
static std::vector<double> path_linear_sgd_vec(const BidirectedGraph &g, const FlatGraph &f, const PathSGDParams &p,
                                               const HipOptions &opt, gfs_stats *stats) {
    std::vector<double> x;
    if (g.node_count() == 0) return x;                               // sgd.rs:242-244
    gfs_graph_view v = f.view();
    gfs_sgd_params cp = p.to_c();
    x.resize(f.node_len.size());
    gfs_stats st;
    int rc = gfs_path_linear_sgd(&v, &cp, &opt.cfg, nullptr, nullptr, 1, x.data(), &st);
    check(rc);
    if (stats) *stats = st;
    if (rc == GFS_NOTHING_TO_DO) {
        std::cerr << "[path_sgd] No paths with multiple steps found\n";   // sgd.rs:259
        x.clear();
    }
    return x;
}

std::unordered_map<size_t, double> path_linear_sgd(const BidirectedGraph &g, const PathSGDParams &p,
                                                   const HipOptions &opt, gfs_stats *stats) {
    std::unordered_map<size_t, double> positions;
    FlatGraph f = g.flatten();
    std::vector<double> x = path_linear_sgd_vec(g, f, p, opt, stats);
    positions.reserve(x.size());
    for (size_t i = 0; i < x.size(); ++i) positions.emplace(i, x[i]);       // sgd.rs:604-607
    return positions;
}

namespace {
struct Lap {                                                        // GFS_TIMING=1: phase times on stderr
    bool on = std::getenv("GFS_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[sgd_sort] %-14s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
}  // namespace

std::vector<Handle> path_sgd_sort(const BidirectedGraph &g, const PathSGDParams &p, const HipOptions &opt,
                                  gfs_stats *stats) {
    // same result as sorting path_linear_sgd's map (sgd.rs:641-672); the sort runs on the device
    std::vector<Handle> out;
    if (g.node_count() == 0) return out;                             // sgd.rs:242-244
    Lap lap;
    FlatGraph f = g.flatten();
    lap("flatten");
    gfs_graph_view v = f.view();
    gfs_sgd_params cp = p.to_c();
    std::vector<double> x(f.node_len.size());
    std::vector<uint64_t> order(x.size());
    gfs_stats st;
    int rc = gfs_path_sgd_sort(&v, &cp, &opt.cfg, nullptr, nullptr, 1, x.data(), order.data(), &st);
    lap("gfs_path_sgd_sort");
    check(rc);
    if (stats) *stats = st;
    if (rc == GFS_NOTHING_TO_DO) {
        std::cerr << "[path_sgd] No paths with multiple steps found\n";   // sgd.rs:259
        return out;
    }
    out.reserve(order.size());
    for (uint64_t idx : order) out.push_back(Handle::forward(f.node_ids[idx]));   // idx -> handle, sgd.rs:649-662
    lap("handles");
    return out;
}

void sgd_sort_only(BidirectedGraph &g, const PathSGDParams &p, uint8_t verbose, const HipOptions &opt, gfs_stats *stats) {
    if (verbose >= 2) std::cerr << "[path_sgd] Starting path-guided SGD\n";
    auto ordering = path_sgd_sort(g, p, opt, stats);
    Lap lap;
    g.apply_ordering(ordering);
    lap("apply_ordering");
    if (verbose >= 2) std::cerr << "[path_sgd] Complete\n";
}

static uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

std::vector<double> default_layout_init(const FlatGraph &f, size_t D, uint64_t seed) {
    // the reference's own start (sgd.rs:829-853): dimension 0 = bp prefix, dimensions >= 1 = StandardNormal * sqrt(2N)
    // from one Xoshiro256+ seeded `seed` (gfs_init_layout restates rand_distr's ziggurat: parity unpinned)
    const size_t N = f.node_len.size(), n = N * 2 * D;
    std::vector<double> c(n, 0.0);
    gfs_graph_view v = f.view();
    gfs_init_layout(&v, D, seed, c.data());
    return c;
}

Layout path_linear_sgd_layout(const BidirectedGraph &g, const LayoutSGDParams &p, const HipOptions &opt, gfs_stats *stats) {
    size_t N = g.node_count();
    if (N == 0) return Layout(p.dimensions, 0);                      // sgd.rs:780-782
    FlatGraph f = g.flatten();
    gfs_graph_view v = f.view();
    gfs_layout_params cp = p.to_c();
    std::vector<double> coords = default_layout_init(f, p.dimensions, p.seed);
    gfs_stats st;
    int rc = gfs_path_linear_sgd_layout(&v, &cp, &opt.cfg, nullptr, nullptr, coords.data(), &st);
    check(rc);
    if (stats) *stats = st;
    if (rc == GFS_NOTHING_TO_DO) {
        std::cerr << "[path_sgd_layout] No paths with multiple steps found\n";   // sgd.rs:796
        return Layout(p.dimensions, N);
    }
    Layout l(p.dimensions, N);
    l.coords = std::move(coords);
    return l;
}

// ---- calculate_layout_stress (sgd.rs:1196-1283) -----------------------------------------------------
namespace {
struct Xo {
    uint64_t s[4];
    explicit Xo(uint64_t seed) { uint64_t sm = seed; for (auto &v : s) v = splitmix64(sm); }
    uint64_t next() {
        uint64_t r = s[0] + s[3], t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t;
        s[3] = (s[3] << 45) | (s[3] >> 19);
        return r;
    }
    uint64_t uniform(uint64_t n) {                                   // rand 0.9 Uniform<usize>(0,n)
        if (n <= 0xFFFFFFFFull) {
            uint32_t range = (uint32_t)n, thresh = (uint32_t)(0u - range) % range;
            for (;;) { uint64_t m = (uint64_t)(uint32_t)(next() >> 32) * range; if ((uint32_t)m >= thresh) return m >> 32; }
        }
        uint64_t thresh = (0ull - n) % n;
        for (;;) { unsigned __int128 m = (unsigned __int128)next() * n; if ((uint64_t)m >= thresh) return (uint64_t)(m >> 64); }
    }
};
}  // namespace

double calculate_layout_stress(const BidirectedGraph &g, const Layout &layout, size_t sample_count) {
    FlatGraph f = g.flatten();
    const size_t S = f.step_node.size();
    if (S < 2) return 0.0;
    std::vector<uint64_t> pos(S); std::vector<uint32_t> pth(S);
    for (size_t p = 0; p + 1 < f.path_first_step.size(); ++p) {
        uint64_t position = 0;
        for (uint64_t s = f.path_first_step[p]; s < f.path_first_step[p + 1]; ++s) {
            pos[s] = position; pth[s] = (uint32_t)p;
            if (f.step_node[s] != GFS_NO_NODE) position += f.node_len[f.step_node[s]];
        }
    }
    Xo rng(12345);
    double sum = 0.0; uint64_t count = 0;
    for (size_t k = 0; k < sample_count; ++k) {
        uint64_t a = rng.uniform(S);
        uint64_t p = pth[a], first = f.path_first_step[p], cnt = f.path_first_step[p + 1] - first;
        if (cnt < 2) continue;
        uint64_t ra = a - first, rb = rng.uniform(cnt);
        if (ra == rb) continue;
        uint64_t sa = first + ra, sb = first + rb;
        double path_dist = std::fabs((double)pos[sa] - (double)pos[sb]);
        if (path_dist == 0.0) continue;
        uint32_t ia = f.step_node[sa], ib = f.step_node[sb];
        if (ia == GFS_NO_NODE || ib == GFS_NO_NODE) continue;
        double err = layout.distance(ia, 0, ib, 0) - path_dist;
        sum += (err * err) / (path_dist * path_dist);
        ++count;
    }
    return count ? std::sqrt(sum / (double)count) : 0.0;
}

}  // namespace gfasort
