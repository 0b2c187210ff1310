// sgd.hpp — C++ host mirror of the reference's SGD entry points on top of libgfasort_hip.so.
//   PathSGDParams        src/sgd.rs:196-234       YgsParams        src/ygs.rs:16-93
//   LayoutSGDParams      src/sgd.rs:676-763       Layout           src/layout.rs:17-163
//   path_linear_sgd      src/sgd.rs:237           path_sgd_sort    src/sgd.rs:641
//   sgd_sort_only        src/ygs.rs:195           path_linear_sgd_layout  src/sgd.rs:773
//   calculate_layout_stress  src/sgd.rs:1196
// Same names, argument meaning and empty-result behaviour as the Rust functions.
#pragma once
#include <cstdint>
#include <iosfwd>
#include <string>
#include <unordered_map>
#include <vector>

#include "graph.hpp"

namespace gfasort {

struct PathSGDParams {                     // defaults: sgd.rs:214-234
    uint64_t iter_max = 100;
    uint64_t iter_with_max_learning_rate = 0;
    uint64_t min_term_updates = 100;
    double delta = 0.0;
    double eps = 0.01;
    double eta_max = 100.0;
    double theta = 0.99;
    uint64_t space = 100;
    uint64_t space_max = 100;
    uint64_t space_quantization_step = 100;
    double cooling_start = 0.5;
    size_t nthreads = 1;
    bool progress = false;
    uint64_t seed = 9399220;
    gfs_sgd_params to_c() const;
};

struct LayoutSGDParams {                   // defaults: sgd.rs:709-729
    size_t dimensions = 2;
    uint64_t iter_max = 30;
    uint64_t iter_with_max_learning_rate = 0;
    uint64_t min_term_updates = 100;
    double delta = 0.0;
    double eps = 0.01;
    double eta_max = 100.0;
    double theta = 0.99;
    uint64_t space = 100;
    uint64_t space_max = 1000;
    uint64_t space_quantization_step = 100;
    double cooling_start = 0.5;
    size_t nthreads = 1;
    bool progress = false;
    uint64_t seed = 9399220;
    static LayoutSGDParams from_graph(const BidirectedGraph &g, size_t dimensions, size_t nthreads);  // sgd.rs:733
    gfs_layout_params to_c() const;
};

struct YgsParams {                         // ygs.rs:16-45
    PathSGDParams path_sgd;
    uint8_t verbose = 0;
    YgsParams();
    static YgsParams from_graph(const BidirectedGraph &g, uint8_t verbose, size_t nthreads);          // ygs.rs:50
};

struct Layout {                            // layout.rs:17-24
    size_t dimensions = 0;
    size_t num_nodes = 0;
    std::vector<double> coords;            // coords[node*2*D + end*D + dim]
    Layout() = default;
    Layout(size_t dims, size_t n) : dimensions(dims), num_nodes(n), coords(n * 2 * dims, 0.0) {}
    static Layout from_vectors(const std::vector<std::vector<double>> &coord_vecs);                   // layout.rs:39
    size_t index(size_t node, size_t end, size_t dim) const { return node * 2 * dimensions + end * dimensions + dim; }
    double get(size_t node, size_t end, size_t dim) const { return coords[index(node, end, dim)]; }
    void set(size_t node, size_t end, size_t dim, double v) { coords[index(node, end, dim)] = v; }
    double distance(size_t na, size_t ea, size_t nb, size_t eb) const;                                // layout.rs:126
    void write_tsv(std::ostream &out) const;                                                          // layout.rs:138
};

std::string rust_display_f64(double v);    // Rust `{}` on f64

// Device launch shape for the HIP engine (no reference equivalent; all-zero = defaults).
struct HipOptions {
    gfs_launch_config cfg{};
};

// sgd.rs:237 — dense index (position in node_order) -> final position; EMPTY when the reference
// returns an empty map.  Throws std::runtime_error on a HIP / argument error.
std::unordered_map<size_t, double> path_linear_sgd(const BidirectedGraph &g, const PathSGDParams &p,
                                                   const HipOptions &opt = {}, gfs_stats *stats = nullptr);
// sgd.rs:641 — handles in ascending position order (ties keep node_order; the reference's tie
// order is HashMap-random).
std::vector<Handle> path_sgd_sort(const BidirectedGraph &g, const PathSGDParams &p, const HipOptions &opt = {},
                                  gfs_stats *stats = nullptr);
// ygs.rs:195 — path_sgd_sort + apply_ordering.
void sgd_sort_only(BidirectedGraph &g, const PathSGDParams &p, uint8_t verbose, const HipOptions &opt = {},
                   gfs_stats *stats = nullptr);
// sgd.rs:773.  Gaussian start of dims >= 1 is drawn here (Box-Muller on SplitMix64(seed); the
// reference's rand_distr ziggurat stream is not reproduced).
Layout path_linear_sgd_layout(const BidirectedGraph &g, const LayoutSGDParams &p, const HipOptions &opt = {},
                              gfs_stats *stats = nullptr);
std::vector<double> default_layout_init(const FlatGraph &f, size_t dims, uint64_t seed);
// sgd.rs:1196 (host, seed 12345)
double calculate_layout_stress(const BidirectedGraph &g, const Layout &layout, size_t sample_count);

}  // namespace gfasort
