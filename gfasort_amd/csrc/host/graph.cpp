// graph.cpp — see graph.hpp.
#include "graph.hpp"

#include <algorithm>
#include <stdexcept>
#include <unordered_map>

namespace gfasort {

void BidirectedGraph::add_node(size_t id, std::string sequence) {
    if (id >= nodes.size()) nodes.resize(id + 1);
    if (!nodes[id].has_value()) node_order.push_back(id);          // only new ids enter node_order
    BiNode n; n.id = id; n.sequence = std::move(sequence);
    nodes[id] = std::move(n);
}

void BidirectedGraph::add_edge(Handle from, Handle to) {
    BiEdge e{from, to};
    BiEdge comp{to.flip(), from.flip()};                           // A+ -> B+  ==  B- -> A-
    if (!edges.count(e) && !edges.count(comp)) edges.insert(e);
}

bool BidirectedGraph::has_edge(Handle from, Handle to) const {
    return edges.count(BiEdge{from, to}) || edges.count(BiEdge{to.flip(), from.flip()});
}

size_t BidirectedGraph::node_count() const {
    size_t n = 0;
    for (const auto &x : nodes) n += x.has_value();
    return n;
}

std::vector<size_t> BidirectedGraph::seeding_order() const {
    if (!node_order.empty()) return node_order;
    std::vector<size_t> ids;
    for (size_t id = 0; id < nodes.size(); ++id) if (nodes[id].has_value()) ids.push_back(id);
    return ids;
}

FlatGraph BidirectedGraph::flatten() const {
    FlatGraph f;
    std::vector<uint32_t> idx_of(nodes.size(), GFS_NO_NODE);
    for (size_t id : seeding_order()) {
        if (id < nodes.size() && nodes[id].has_value()) {          // sgd.rs:286-294
            if (idx_of[id] != GFS_NO_NODE) continue;
            idx_of[id] = (uint32_t)f.node_len.size();
            f.node_len.push_back((uint32_t)nodes[id]->sequence.size());
            f.node_ids.push_back(id);
        }
    }
    f.path_first_step.push_back(0);
    for (const auto &p : paths) {                                  // PathIndex::from_graph order (sgd.rs:41)
        for (Handle h : p.steps) {
            size_t id = h.node_id();
            f.step_node.push_back(id < idx_of.size() ? idx_of[id] : GFS_NO_NODE);
            f.step_is_rev.push_back(h.is_reverse() ? 1 : 0);
        }
        f.path_first_step.push_back(f.step_node.size());
    }
    return f;
}

void BidirectedGraph::apply_ordering(const std::vector<Handle> &ordering) {
    if (ordering.empty()) return;                                  // graph_ops.rs:1940
    std::unordered_map<size_t, size_t> old_to_new;
    for (size_t k = 0; k < ordering.size(); ++k) old_to_new[ordering[k].node_id()] = k + 1;   // 1-based
    size_t max_new = 0;
    for (auto &kv : old_to_new) max_new = std::max(max_new, kv.second);
    std::vector<std::optional<BiNode>> new_nodes(max_new + 1);
    for (auto &kv : old_to_new) {
        if (kv.first < nodes.size() && nodes[kv.first].has_value()) {
            BiNode n = *nodes[kv.first];
            n.id = kv.second;
            n.rank = (uint64_t)(kv.second - 1);
            new_nodes[kv.second] = std::move(n);
        }
    }
    nodes = std::move(new_nodes);
    std::unordered_set<BiEdge, BiEdgeHash> new_edges;
    for (const auto &e : edges) {
        auto f = old_to_new.find(e.from.node_id()), t = old_to_new.find(e.to.node_id());
        if (f != old_to_new.end() && t != old_to_new.end())
            new_edges.insert(BiEdge{Handle::make(f->second, e.from.is_reverse()), Handle::make(t->second, e.to.is_reverse())});
    }
    edges = std::move(new_edges);
    for (auto &p : paths)
        for (auto &h : p.steps) {
            auto it = old_to_new.find(h.node_id());
            if (it != old_to_new.end()) h = Handle::make(it->second, h.is_reverse());
        }
    // node_order is NOT updated — exactly like the reference (it is written only by add_node).
}

void BidirectedGraph::write_gfa(std::ostream &out) const {
    out << "H\tVN:Z:1.0\n";
    for (size_t id = 0; id < nodes.size(); ++id)
        if (nodes[id].has_value()) out << "S\t" << id << "\t" << nodes[id]->sequence << "\n";
    std::vector<BiEdge> es(edges.begin(), edges.end());
    std::sort(es.begin(), es.end(), [](const BiEdge &a, const BiEdge &b) {
        return a.from.v != b.from.v ? a.from.v < b.from.v : a.to.v < b.to.v;
    });
    for (const auto &e : es)
        out << "L\t" << e.from.node_id() << "\t" << e.from.orientation_char() << "\t" << e.to.node_id() << "\t"
            << e.to.orientation_char() << "\t0M\n";
    for (const auto &p : paths) {
        out << "P\t" << p.name << "\t";
        for (size_t k = 0; k < p.steps.size(); ++k) {
            if (k) out << ",";
            out << p.steps[k].node_id() << p.steps[k].orientation_char();
        }
        out << "\t*\n";
    }
}

// ---- parse_gfa ---------------------------------------------------------------------------------
static std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out;
    size_t b = 0;
    for (;;) {
        size_t e = s.find(sep, b);
        if (e == std::string::npos) { out.push_back(s.substr(b)); break; }
        out.push_back(s.substr(b, e - b));
        b = e + 1;
    }
    return out;
}

static size_t parse_usize(const std::string &s, const char *what) {
    // Rust `str::parse::<usize>`: optional '+', digits only, no whitespace, no overflow
    size_t i = 0;
    if (!s.empty() && s[0] == '+') i = 1;
    if (i >= s.size()) throw std::runtime_error(std::string("Failed to parse ") + what + ": cannot parse integer from empty string");
    uint64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') throw std::runtime_error(std::string("Failed to parse ") + what + ": invalid digit found in string");
        uint64_t nv = v * 10 + (uint64_t)(s[i] - '0');
        if (nv / 10 != v) throw std::runtime_error(std::string("Failed to parse ") + what + ": number too large to fit in target type");
        v = nv;
    }
    return (size_t)v;
}

static std::string trim(const std::string &s) {
    size_t b = 0, e = s.size();
    while (b < e && isspace((unsigned char)s[b])) ++b;
    while (e > b && isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}

BidirectedGraph parse_gfa(const std::string &content) {
    BidirectedGraph g;
    std::vector<std::string> lines;
    {
        size_t b = 0;
        while (b <= content.size()) {
            size_t e = content.find('\n', b);
            std::string ln = content.substr(b, e == std::string::npos ? std::string::npos : e - b);
            if (!ln.empty() && ln.back() == '\r') ln.pop_back();   // str::lines strips \r\n
            if (e == std::string::npos) { if (!ln.empty()) lines.push_back(ln); break; }
            lines.push_back(ln);
            b = e + 1;
        }
    }
    for (const auto &ln : lines)                                   // pass 1: S
        if (!ln.empty() && ln[0] == 'S') {
            auto parts = split(ln, '\t');
            if (parts.size() >= 3) g.add_node(parse_usize(parts[1], "node ID"), parts[2]);
        }
    for (const auto &ln : lines)                                   // pass 2: L
        if (!ln.empty() && ln[0] == 'L') {
            auto parts = split(ln, '\t');
            if (parts.size() >= 5) {
                size_t f = parse_usize(parts[1], "from ID"), t = parse_usize(parts[3], "to ID");
                g.add_edge(parts[2] == "+" ? Handle::forward(f) : Handle::reverse(f),
                           parts[4] == "+" ? Handle::forward(t) : Handle::reverse(t));
            }
        }
    for (const auto &ln : lines)                                   // pass 3: P
        if (!ln.empty() && ln[0] == 'P') {
            auto parts = split(ln, '\t');
            if (parts.size() >= 3) {
                BiPath p; p.name = parts[1];
                for (const auto &raw : split(parts[2], ',')) {
                    std::string s = trim(raw);
                    if (s.empty()) continue;
                    char orient = s.back();
                    size_t id = parse_usize(s.substr(0, s.size() - 1), "path node ID");
                    p.steps.push_back(orient == '+' ? Handle::forward(id) : Handle::reverse(id));
                }
                g.paths.push_back(std::move(p));
            }
        }
    return g;
}

}  // namespace gfasort
