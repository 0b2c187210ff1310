// graph.hpp — C++ host-side mirror of the reference's graph model, as far as the SGD path
// (`Y`, `L`) touches it.  Same names and semantics as the Rust types:
//   Handle            src/graph.rs:9-64      (u64, LSB = orientation)
//   BiNode/BiPath/BiEdge  src/graph.rs:87-199
//   BidirectedGraph   src/graph_ops.rs:10-16 (+ add_node :613, add_edge :626, node_count :535,
//                     write_gfa :693, apply_ordering :1939)
//   parse_gfa         src/bin/gfasort.rs:88-167 (CLI reader: numeric ids kept)
// plus the flattened SoA mirror (gfs_graph_view) that the HIP library consumes.
#pragma once
#include <cstdint>
#include <functional>
#include <optional>
#include <ostream>
#include <string>
#include <iterator>
#include <vector>

#include "../../../include/gfasort_hip.h"

namespace gfasort {

struct Handle {
    uint64_t v = 0;
    static Handle make(size_t node_id, bool is_reverse) { return Handle{((uint64_t)node_id << 1) | (is_reverse ? 1u : 0u)}; }
    static Handle forward(size_t node_id) { return make(node_id, false); }
    static Handle reverse(size_t node_id) { return make(node_id, true); }
    static Handle from_u64(uint64_t x) { return Handle{x}; }
    size_t node_id() const { return (size_t)(v >> 1); }
    bool is_reverse() const { return (v & 1) == 1; }
    char orientation_char() const { return is_reverse() ? '-' : '+'; }
    Handle flip() const { return Handle{v ^ 1}; }
    uint64_t as_u64() const { return v; }
    bool operator==(const Handle &o) const { return v == o.v; }
    bool operator<(const Handle &o) const { return v < o.v; }
    std::string to_string() const { return std::to_string(node_id()) + orientation_char(); }
};

struct BiNode {
    size_t id = 0;
    std::string sequence;
    std::optional<uint64_t> rank;
};

struct BiPath {
    std::string name;
    std::vector<Handle> steps;
    void add_step(Handle h) { steps.push_back(h); }
};

struct BiEdge {
    Handle from, to;
    bool operator==(const BiEdge &o) const { return from == o.from && to == o.to; }
};
struct BiEdgeHash {
    size_t operator()(const BiEdge &e) const {
        uint64_t x = e.from.v * 0x9E3779B97F4A7C15ull ^ (e.to.v + 0x7F4A7C15ull + (e.from.v << 6));
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        return (size_t)x;
    }
};

// The reference keeps edges in a HashSet<BiEdge> (graph_ops.rs:12).  Same operations, stored as an
// open-addressing table of 16-byte keys: a million-edge graph is rebuilt once by the parser and once
// by apply_ordering, and node-based buckets made those the longest host phases.
class EdgeSet {
public:
    class const_iterator {
    public:
        const_iterator(const EdgeSet *s, size_t i) : s_(s), i_(i) { skip(); }
        const BiEdge &operator*() const { return s_->slots_[i_]; }
        const BiEdge *operator->() const { return &s_->slots_[i_]; }
        const_iterator &operator++() { ++i_; skip(); return *this; }
        bool operator!=(const const_iterator &o) const { return i_ != o.i_; }
        bool operator==(const const_iterator &o) const { return i_ == o.i_; }
        using iterator_category = std::forward_iterator_tag;
        using value_type = BiEdge; using difference_type = std::ptrdiff_t;
        using pointer = const BiEdge *; using reference = const BiEdge &;
    private:
        void skip() { while (i_ < s_->used_.size() && !s_->used_[i_]) ++i_; }
        const EdgeSet *s_; size_t i_;
    };
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    void clear() { slots_.clear(); used_.clear(); n_ = 0; }
    void reserve(size_t n) { if (n * 10 > slots_.size() * 7) rehash(n); }
    size_t count(const BiEdge &e) const {
        if (slots_.empty()) return 0;
        const size_t mask = slots_.size() - 1;
        for (size_t i = BiEdgeHash()(e) & mask;; i = (i + 1) & mask) {
            if (!used_[i]) return 0;
            if (slots_[i] == e) return 1;
        }
    }
    bool insert(const BiEdge &e) {                              // true when newly inserted
        if ((n_ + 1) * 10 > slots_.size() * 7) rehash(n_ + 1);
        const size_t mask = slots_.size() - 1;
        for (size_t i = BiEdgeHash()(e) & mask;; i = (i + 1) & mask) {
            if (!used_[i]) { slots_[i] = e; used_[i] = 1; ++n_; return true; }
            if (slots_[i] == e) return false;
        }
    }
    void prefetch(const BiEdge &e) const {                      // ahead of count/insert on a table that will not grow
        if (slots_.empty()) return;
        const size_t i = BiEdgeHash()(e) & (slots_.size() - 1);
        __builtin_prefetch(&slots_[i]); __builtin_prefetch(&used_[i]);
    }
    const_iterator begin() const { return const_iterator(this, 0); }
    const_iterator end() const { return const_iterator(this, used_.size()); }
private:
    void rehash(size_t want) {
        size_t cap = 16;
        while (cap * 7 < want * 10) cap <<= 1;
        if (cap < slots_.size()) cap = slots_.size();
        std::vector<BiEdge> old_s(cap); std::vector<uint8_t> old_u(cap, 0);
        old_s.swap(slots_); old_u.swap(used_);
        n_ = 0;
        for (size_t i = 0; i < old_u.size(); ++i) if (old_u[i]) insert(old_s[i]);
    }
    std::vector<BiEdge> slots_;
    std::vector<uint8_t> used_;
    size_t n_ = 0;
};

// Flattened, owning SoA mirror of what src/sgd.rs reads (the "device mirror" source).
struct FlatGraph {
    std::vector<uint32_t> node_len;        // by dense index (node_order order)
    std::vector<uint32_t> step_node;       // dense index or GFS_NO_NODE
    std::vector<uint8_t> step_is_rev;
    std::vector<uint64_t> path_first_step; // n_paths + 1
    std::vector<size_t> node_ids;          // dense index -> node id (node_order)
    gfs_graph_view view() const {
        gfs_graph_view v;
        v.n_nodes = node_len.size(); v.n_steps = step_node.size(); v.n_paths = path_first_step.size() - 1;
        v.node_len = node_len.data(); v.step_node = step_node.data();
        v.step_is_rev = step_is_rev.data(); v.path_first_step = path_first_step.data();
        return v;
    }
};

struct BidirectedGraph {
    std::vector<std::optional<BiNode>> nodes;                 // index = node id
    EdgeSet edges;                                            // HashSet<BiEdge>
    std::vector<BiPath> paths;
    std::vector<size_t> node_order;                           // order of add_node calls (GFA S lines)

    void add_node(size_t id, std::string sequence);           // graph_ops.rs:613-623
    void add_edge(Handle from, Handle to);                    // graph_ops.rs:626-637
    void add_edges(const std::vector<BiEdge> &list);          // add_edge for each, in order (table sized once)
    bool has_edge(Handle from, Handle to) const;              // graph_ops.rs:648-652
    size_t node_count() const;                                // graph_ops.rs:535-537
    void apply_ordering(const std::vector<Handle> &ordering); // graph_ops.rs:1939-2025
    void write_gfa(std::ostream &out) const;                  // graph_ops.rs:693-738 (L lines sorted; the
                                                              // reference iterates a HashSet: random order)
    // node ids in the order sgd.rs seeds positions: node_order, else sorted ids (sgd.rs:276-284)
    std::vector<size_t> seeding_order() const;
    FlatGraph flatten() const;                                // PathIndex inputs + handle_to_idx (sgd.rs:286-294)
};

// Host threads for the text and relabelling passes over path steps (parse, flatten, apply_ordering,
// write): 0 = auto (the CPUs this process may run on, at most 16).  Results do not depend on it.
void set_io_threads(size_t n);
size_t io_threads();
// fn(0..n-1) on the host threads; the exception of the lowest failing item is rethrown
void parallel_for(size_t n, const std::function<void(size_t)> &fn);

// Whole file into a string (one read of the file's size).  Throws std::runtime_error(strerror).
std::string read_file(const std::string &path);

// CLI-compatible reader (src/bin/gfasort.rs:88-167).  Throws std::runtime_error with the
// reference's messages ("Failed to parse node ID: ...").
BidirectedGraph parse_gfa(const std::string &content);

}  // namespace gfasort
