// graph.cpp — see graph.hpp.
#include "graph.hpp"

#include <algorithm>
#include <cstring>
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
    // old id -> new 1-based id (graph_ops.rs:1953-1957); a dense table instead of the reference's
    // HashMap: ids outside it, or with 0, are "not in the ordering"
    size_t max_old = nodes.size();
    for (Handle h : ordering) max_old = std::max(max_old, h.node_id() + 1);
    std::vector<size_t> old_to_new(max_old, 0);
    for (size_t k = 0; k < ordering.size(); ++k) old_to_new[ordering[k].node_id()] = k + 1;
    auto lookup = [&](size_t id) -> size_t { return id < old_to_new.size() ? old_to_new[id] : 0; };
    size_t max_new = 0;
    for (size_t v : old_to_new) max_new = std::max(max_new, v);
    std::vector<std::optional<BiNode>> new_nodes(max_new + 1);
    for (size_t id = 0; id < nodes.size(); ++id) {
        size_t nid = lookup(id);
        if (nid && nodes[id].has_value()) {
            BiNode n = std::move(*nodes[id]);
            n.id = nid;
            n.rank = (uint64_t)(nid - 1);                          // 0-based rank
            new_nodes[nid] = std::move(n);
        }
    }
    nodes = std::move(new_nodes);
    std::unordered_set<BiEdge, BiEdgeHash> new_edges;
    new_edges.reserve(edges.size());
    for (const auto &e : edges) {
        size_t f = lookup(e.from.node_id()), t = lookup(e.to.node_id());
        if (f && t) new_edges.insert(BiEdge{Handle::make(f, e.from.is_reverse()), Handle::make(t, e.to.is_reverse())});
    }
    edges = std::move(new_edges);
    for (auto &p : paths)
        for (auto &h : p.steps) {
            size_t nid = lookup(h.node_id());
            if (nid) h = Handle::make(nid, h.is_reverse());
        }
    // node_order is NOT updated — exactly like the reference (it is written only by add_node).
}

namespace {
inline void put_uint(std::string &b, uint64_t v) {
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) b.push_back(tmp[--n]);
}
}  // namespace

void BidirectedGraph::write_gfa(std::ostream &out) const {
    // one growing buffer, flushed in 8 MB pieces (the step lists of long paths are the bulk)
    std::string b;
    b.reserve(9u << 20);
    auto flush = [&](bool force) { if (force || b.size() > (8u << 20)) { out.write(b.data(), (std::streamsize)b.size()); b.clear(); } };
    b += "H\tVN:Z:1.0\n";
    for (size_t id = 0; id < nodes.size(); ++id)
        if (nodes[id].has_value()) {
            b += "S\t"; put_uint(b, id); b.push_back('\t'); b += nodes[id]->sequence; b.push_back('\n');
            flush(false);
        }
    std::vector<BiEdge> es(edges.begin(), edges.end());
    std::sort(es.begin(), es.end(), [](const BiEdge &a, const BiEdge &c) {
        return a.from.v != c.from.v ? a.from.v < c.from.v : a.to.v < c.to.v;
    });
    for (const auto &e : es) {
        b += "L\t"; put_uint(b, e.from.node_id()); b.push_back('\t'); b.push_back(e.from.orientation_char());
        b.push_back('\t'); put_uint(b, e.to.node_id()); b.push_back('\t'); b.push_back(e.to.orientation_char());
        b += "\t0M\n";
        flush(false);
    }
    for (const auto &p : paths) {
        b += "P\t"; b += p.name; b.push_back('\t');
        for (size_t k = 0; k < p.steps.size(); ++k) {
            if (k) b.push_back(',');
            put_uint(b, p.steps[k].node_id()); b.push_back(p.steps[k].orientation_char());
            if ((k & 0xFFFF) == 0) flush(false);
        }
        b += "\t*\n";
        flush(false);
    }
    flush(true);
}

// ---- parse_gfa ---------------------------------------------------------------------------------
// Field access without allocation: the k-th tab-separated field of a line.
namespace {
struct Line { const char *b, *e; };
inline bool field(const Line &ln, int k, const char *&fb, const char *&fe) {
    const char *p = ln.b;
    for (int i = 0; i < k; ++i) {
        const char *t = (const char *)memchr(p, '\t', (size_t)(ln.e - p));
        if (!t) return false;
        p = t + 1;
    }
    const char *t = (const char *)memchr(p, '\t', (size_t)(ln.e - p));
    fb = p; fe = t ? t : ln.e;
    return true;
}
inline int count_fields(const Line &ln) {
    int n = 1;
    for (const char *p = ln.b; (p = (const char *)memchr(p, '\t', (size_t)(ln.e - p))); ++p) ++n;
    return n;
}
size_t parse_usize(const char *b, const char *e, const char *what) {
    // Rust `str::parse::<usize>`: optional '+', digits only, no whitespace, no overflow
    if (b < e && *b == '+') ++b;
    if (b >= e) throw std::runtime_error(std::string("Failed to parse ") + what + ": cannot parse integer from empty string");
    uint64_t v = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') throw std::runtime_error(std::string("Failed to parse ") + what + ": invalid digit found in string");
        uint64_t nv = v * 10 + (uint64_t)(*b - '0');
        if (nv / 10 != v) throw std::runtime_error(std::string("Failed to parse ") + what + ": number too large to fit in target type");
        v = nv;
    }
    return (size_t)v;
}
}  // namespace

BidirectedGraph parse_gfa(const std::string &content) {
    BidirectedGraph g;
    // str::lines(): split on '\n', strip one trailing '\r', no final empty line
    std::vector<Line> lines;
    {
        const char *p = content.data(), *end = p + content.size();
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? nl : end;
            Line ln{p, (le > p && le[-1] == '\r') ? le - 1 : le};
            lines.push_back(ln);
            if (!nl) break;
            p = nl + 1;
        }
    }
    const char *fb = nullptr, *fe = nullptr;
    for (const Line &ln : lines)                                   // pass 1: S (gfasort.rs:93-103)
        if (ln.b < ln.e && *ln.b == 'S' && count_fields(ln) >= 3) {
            field(ln, 1, fb, fe);
            size_t id = parse_usize(fb, fe, "node ID");
            field(ln, 2, fb, fe);
            g.add_node(id, std::string(fb, fe));
        }
    for (const Line &ln : lines)                                   // pass 2: L (gfasort.rs:106-132)
        if (ln.b < ln.e && *ln.b == 'L' && count_fields(ln) >= 5) {
            field(ln, 1, fb, fe); size_t f = parse_usize(fb, fe, "from ID");
            field(ln, 2, fb, fe); bool ff = (fe - fb == 1 && *fb == '+');
            field(ln, 3, fb, fe); size_t t = parse_usize(fb, fe, "to ID");
            field(ln, 4, fb, fe); bool tf = (fe - fb == 1 && *fb == '+');
            g.add_edge(ff ? Handle::forward(f) : Handle::reverse(f), tf ? Handle::forward(t) : Handle::reverse(t));
        }
    for (const Line &ln : lines)                                   // pass 3: P (gfasort.rs:135-163)
        if (ln.b < ln.e && *ln.b == 'P' && count_fields(ln) >= 3) {
            BiPath p;
            field(ln, 1, fb, fe); p.name.assign(fb, fe);
            field(ln, 2, fb, fe);
            const char *q = fb;
            while (q <= fe) {
                const char *c = (const char *)memchr(q, ',', (size_t)(fe - q));
                const char *sb = q, *se = c ? c : fe;
                while (sb < se && isspace((unsigned char)*sb)) ++sb;   // step_str.trim()
                while (se > sb && isspace((unsigned char)se[-1])) --se;
                if (sb < se) {
                    char orient = se[-1];
                    size_t id = parse_usize(sb, se - 1, "path node ID");
                    p.steps.push_back(orient == '+' ? Handle::forward(id) : Handle::reverse(id));
                }
                if (!c) break;
                q = c + 1;
            }
            g.paths.push_back(std::move(p));
        }
    return g;
}

}  // namespace gfasort
