// graph.cpp — see graph.hpp.
#include "graph.hpp"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <stdexcept>
#include <thread>

#include <sched.h>

namespace gfasort {

// ---- host threads ----------------------------------------------------------------------------------
namespace {
std::atomic<size_t> g_io_threads{0};

// items 0..n-1 handed out one at a time; the exception of the LOWEST failing item is rethrown, which is
// the one a sequential pass would have hit first
void parallel_for_impl(size_t n, const std::function<void(size_t)> &fn) {
    size_t nt = std::min(io_threads(), n);
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::exception_ptr> err(n);
    std::atomic<bool> failed{false};
    auto worker = [&] {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= n) return;
            try { fn(i); } catch (...) { err[i] = std::current_exception(); failed = true; }
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; ++t) th.emplace_back(worker);
    worker();
    for (auto &t : th) t.join();
    if (failed) for (auto &e : err) if (e) std::rethrow_exception(e);
}

// (path, [begin, end)) pieces of at most `chunk` steps covering every path in order
struct StepChunk { size_t path, begin, end; };
std::vector<StepChunk> step_chunks(const std::vector<BiPath> &paths, size_t chunk) {
    std::vector<StepChunk> out;
    for (size_t p = 0; p < paths.size(); ++p)
        for (size_t b = 0; b < paths[p].steps.size(); b += chunk)
            out.push_back({p, b, std::min(paths[p].steps.size(), b + chunk)});
    return out;
}
}  // namespace

void parallel_for(size_t n, const std::function<void(size_t)> &fn) { parallel_for_impl(n, fn); }
void set_io_threads(size_t n) { g_io_threads = n; }
size_t io_threads() {
    size_t n = g_io_threads;
    if (n) return n;
    if (const char *env = std::getenv("GFS_IO_THREADS")) { long v = std::atol(env); if (v > 0) return (size_t)v; }
    cpu_set_t set;
    size_t avail = 1;
    if (sched_getaffinity(0, sizeof set, &set) == 0) avail = (size_t)CPU_COUNT(&set);
    else if (std::thread::hardware_concurrency()) avail = std::thread::hardware_concurrency();
    return std::max<size_t>(1, std::min<size_t>(avail, 16));
}

std::string read_file(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error(std::strerror(errno));
    std::string content;
    if (std::fseek(f, 0, SEEK_END) == 0) {
        long sz = std::ftell(f);
        if (sz > 0) content.resize((size_t)sz);
        std::rewind(f);
    }
    size_t got = content.empty() ? 0 : std::fread(&content[0], 1, content.size(), f);
    content.resize(got);
    char tail[1 << 16];                                            // not seekable, or grew meanwhile
    for (size_t n; (n = std::fread(tail, 1, sizeof tail, f)) > 0;) content.append(tail, n);
    bool bad = std::ferror(f);
    int e = errno;
    std::fclose(f);
    if (bad) throw std::runtime_error(std::strerror(e));
    return content;
}

void BidirectedGraph::add_node(size_t id, std::string sequence) {
    if (id >= nodes.size()) nodes.resize(id + 1);
    if (!nodes[id].has_value()) node_order.push_back(id);          // only new ids enter node_order
    BiNode n; n.id = id; n.sequence = std::move(sequence);
    nodes[id] = std::move(n);
}

void BidirectedGraph::add_edge(Handle from, Handle to) {
    BiEdge e{from, to};
    BiEdge comp{to.flip(), from.flip()};                           // A+ -> B+  ==  B- -> A-
    if (!edges.count(comp)) edges.insert(e);                       // insert is a no-op when e is present
}

void BidirectedGraph::add_edges(const std::vector<BiEdge> &list) {
    edges.reserve(edges.size() + list.size());
    constexpr size_t AHEAD = 16;                                   // the table is far larger than the caches
    for (size_t k = 0; k < list.size(); ++k) {
        if (k + AHEAD < list.size()) {
            const BiEdge &n = list[k + AHEAD];
            edges.prefetch(n); edges.prefetch(BiEdge{n.to.flip(), n.from.flip()});
        }
        add_edge(list[k].from, list[k].to);
    }
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
    for (const auto &p : paths)                                    // PathIndex::from_graph order (sgd.rs:41)
        f.path_first_step.push_back(f.path_first_step.back() + p.steps.size());
    f.step_node.resize(f.path_first_step.back());
    f.step_is_rev.resize(f.path_first_step.back());
    const auto chunks = step_chunks(paths, (size_t)1 << 18);
    parallel_for(chunks.size(), [&](size_t c) {
        const StepChunk &ck = chunks[c];
        const Handle *st = paths[ck.path].steps.data();
        uint32_t *sn = f.step_node.data() + f.path_first_step[ck.path];
        uint8_t *sr = f.step_is_rev.data() + f.path_first_step[ck.path];
        for (size_t k = ck.begin; k < ck.end; ++k) {
            size_t id = st[k].node_id();
            sn[k] = id < idx_of.size() ? idx_of[id] : GFS_NO_NODE;
            sr[k] = st[k].is_reverse() ? 1 : 0;
        }
    });
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
    {
        // an ordering names a node at most once in practice; when it does not, the last occurrence's slot is the one
        // old_to_new holds, so every old node still has exactly one target and the moves are independent
        const size_t BLOCK = (size_t)1 << 16;
        parallel_for((nodes.size() + BLOCK - 1) / BLOCK, [&](size_t b) {
            for (size_t id = b * BLOCK; id < std::min(nodes.size(), (b + 1) * BLOCK); ++id) {
                size_t nid = lookup(id);
                if (nid && nodes[id].has_value()) {
                    BiNode n = std::move(*nodes[id]);
                    n.id = nid;
                    n.rank = (uint64_t)(nid - 1);                  // 0-based rank
                    new_nodes[nid] = std::move(n);
                }
            }
        });
    }
    nodes = std::move(new_nodes);
    std::vector<BiEdge> relabelled(edges.begin(), edges.end());
    {
        std::vector<uint8_t> keep(relabelled.size(), 0);
        const size_t BLOCK = (size_t)1 << 16;
        parallel_for((relabelled.size() + BLOCK - 1) / BLOCK, [&](size_t b) {
            for (size_t k = b * BLOCK; k < std::min(relabelled.size(), (b + 1) * BLOCK); ++k) {
                const BiEdge e = relabelled[k];
                size_t f = lookup(e.from.node_id()), t = lookup(e.to.node_id());
                if (f && t) { relabelled[k] = BiEdge{Handle::make(f, e.from.is_reverse()), Handle::make(t, e.to.is_reverse())}; keep[k] = 1; }
            }
        });
        size_t n = 0;
        for (size_t k = 0; k < relabelled.size(); ++k) if (keep[k]) relabelled[n++] = relabelled[k];
        relabelled.resize(n);
    }
    EdgeSet new_edges;
    new_edges.reserve(relabelled.size());
    for (size_t k = 0; k < relabelled.size(); ++k) {
        if (k + 16 < relabelled.size()) new_edges.prefetch(relabelled[k + 16]);
        new_edges.insert(relabelled[k]);                           // plain set insert, as graph_ops.rs:1990-2006
    }
    edges = std::move(new_edges);
    const auto chunks = step_chunks(paths, (size_t)1 << 18);
    parallel_for(chunks.size(), [&](size_t c) {
        const StepChunk &ck = chunks[c];
        Handle *st = paths[ck.path].steps.data();
        for (size_t k = ck.begin; k < ck.end; ++k) {
            size_t nid = lookup(st[k].node_id());
            if (nid) st[k] = Handle::make(nid, st[k].is_reverse());
        }
    });
    // node_order is NOT updated — exactly like the reference (it is written only by add_node).
}

namespace {
inline void put_uint(std::string &b, uint64_t v) {
    char tmp[24]; int n = 24;
    do { tmp[--n] = (char)('0' + v % 10); v /= 10; } while (v);
    b.append(tmp + n, (size_t)(24 - n));
}
}  // namespace

namespace {
// items [0, n) in blocks of `block`: fmt(lo, hi, text) appends the block's text (host threads, a bounded window of
// blocks at a time); the blocks are written in order
void write_blocks(std::ostream &out, size_t n, size_t block, const std::function<void(size_t, size_t, std::string &)> &fmt) {
    const size_t nblocks = (n + block - 1) / block;
    const size_t window = std::max<size_t>(io_threads() * 4, 1);
    std::vector<std::string> piece(std::min(window, std::max<size_t>(nblocks, 1)));
    for (size_t base = 0; base < nblocks; base += window) {
        const size_t cnt = std::min(window, nblocks - base);
        parallel_for(cnt, [&](size_t i) {
            piece[i].clear();
            fmt((base + i) * block, std::min(n, (base + i + 1) * block), piece[i]);
        });
        for (size_t i = 0; i < cnt; ++i) out.write(piece[i].data(), (std::streamsize)piece[i].size());
    }
}
}  // namespace

void BidirectedGraph::write_gfa(std::ostream &out) const {
    // one growing buffer, flushed in 8 MB pieces (the step lists of long paths are the bulk)
    const bool timing = std::getenv("GFS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[write_gfa] %-10s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    std::string b;
    b.reserve(9u << 20);
    auto flush = [&](bool force) { if (force || b.size() > (8u << 20)) { out.write(b.data(), (std::streamsize)b.size()); b.clear(); } };
    b += "H\tVN:Z:1.0\n";
    flush(true);
    write_blocks(out, nodes.size(), (size_t)1 << 16, [&](size_t lo, size_t hi, std::string &o) {
        for (size_t id = lo; id < hi; ++id)
            if (nodes[id].has_value()) {
                o += "S\t"; put_uint(o, id); o.push_back('\t'); o += nodes[id]->sequence; o.push_back('\n');
            }
    });
    lap("S");
    std::vector<BiEdge> es(edges.begin(), edges.end());
    lap("L copy");
    {
        auto less = [](const BiEdge &a, const BiEdge &c) { return a.from.v != c.from.v ? a.from.v < c.from.v : a.to.v < c.to.v; };
        // sorted runs by the host threads, then pairwise merges (keys are unique: any order of equal work is the same)
        size_t runs = 1;
        while (runs * 2 <= io_threads() && es.size() / (runs * 2) >= 65536) runs *= 2;
        auto bound = [&](size_t r, size_t of) { return es.size() * r / of; };
        parallel_for(runs, [&](size_t r) { std::sort(es.begin() + (std::ptrdiff_t)bound(r, runs), es.begin() + (std::ptrdiff_t)bound(r + 1, runs), less); });
        for (size_t width = 1; width < runs; width *= 2)
            parallel_for(runs / (2 * width), [&](size_t m) {
                const size_t lo = bound(2 * width * m, runs), mid = bound(2 * width * m + width, runs), hi = bound(2 * width * (m + 1), runs);
                std::inplace_merge(es.begin() + (std::ptrdiff_t)lo, es.begin() + (std::ptrdiff_t)mid, es.begin() + (std::ptrdiff_t)hi, less);
            });
    }
    lap("L sort");
    write_blocks(out, es.size(), (size_t)1 << 16, [&](size_t lo, size_t hi, std::string &o) {
        o.reserve((hi - lo) * 28);
        for (size_t k = lo; k < hi; ++k) {
            const BiEdge &e = es[k];
            o += "L\t"; put_uint(o, e.from.node_id()); o.push_back('\t'); o.push_back(e.from.orientation_char());
            o.push_back('\t'); put_uint(o, e.to.node_id()); o.push_back('\t'); o.push_back(e.to.orientation_char());
            o += "\t0M\n";
        }
    });
    lap("L");
    // step lists: formatted in pieces by the host threads, a bounded window of pieces at a time, written in order
    const auto chunks = step_chunks(paths, (size_t)1 << 17);
    const size_t window = std::max<size_t>(io_threads() * 4, 1);
    std::vector<std::string> piece(std::min(window, chunks.size()));
    size_t next_path = 0;                                          // first path whose header is still to be written
    auto header = [&](size_t p) { b += "P\t"; b += paths[p].name; b.push_back('\t'); };
    auto close_paths_before = [&](size_t p, bool started) {        // empty paths, and the tail of the previous one
        if (started) b += "\t*\n";
        for (; next_path < p; ++next_path) { header(next_path); b += "\t*\n"; }
    };
    bool started = false;
    for (size_t base = 0; base < chunks.size(); base += window) {
        const size_t n = std::min(window, chunks.size() - base);
        parallel_for(n, [&](size_t i) {
            const StepChunk &ck = chunks[base + i];
            const Handle *st = paths[ck.path].steps.data();
            std::string &o = piece[i];
            o.clear();
            o.reserve((ck.end - ck.begin) * 10);
            for (size_t k = ck.begin; k < ck.end; ++k) {
                if (k) o.push_back(',');
                put_uint(o, st[k].node_id()); o.push_back(st[k].orientation_char());
            }
        });
        for (size_t i = 0; i < n; ++i) {
            const StepChunk &ck = chunks[base + i];
            if (ck.begin == 0) {
                close_paths_before(ck.path, started);
                header(ck.path); next_path = ck.path + 1; started = true;
            }
            flush(true);
            out.write(piece[i].data(), (std::streamsize)piece[i].size());
        }
    }
    close_paths_before(paths.size(), started);
    flush(true);
    lap("P");
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
        const uint64_t d = (uint64_t)(*b - '0');
        if (v > 1844674407370955161ull || (v == 1844674407370955161ull && d > 5))     // v*10 + d > u64::MAX
            throw std::runtime_error(std::string("Failed to parse ") + what + ": number too large to fit in target type");
        v = v * 10 + d;
    }
    return (size_t)v;
}
}  // namespace

BidirectedGraph parse_gfa(const std::string &content) {
    BidirectedGraph g;
    const bool timing = std::getenv("GFS_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[parse_gfa] %-10s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // str::lines(): split on '\n', strip one trailing '\r', no final empty line.  Lines are sorted by their
    // first byte into the three lists the reference's three passes would visit (gfasort.rs:93,106,135).
    std::vector<Line> s_lines, l_lines, p_lines;
    {
        // regions of the text that start and end at line boundaries, scanned by the host threads
        const char *base = content.data(), *end = base + content.size();
        const size_t nreg = std::max<size_t>(1, std::min<size_t>(io_threads(), content.size() >> 20));
        std::vector<const char *> cut(nreg + 1, end);
        cut[0] = base;
        for (size_t r = 1; r < nreg; ++r) {
            const char *guess = base + content.size() * r / nreg;
            const char *nl = guess <= cut[r - 1] ? nullptr : (const char *)memchr(guess, '\n', (size_t)(end - guess));
            cut[r] = nl ? nl + 1 : end;
            if (cut[r] < cut[r - 1]) cut[r] = cut[r - 1];
        }
        struct Found { std::vector<Line> s, l, p; };
        std::vector<Found> found(nreg);
        parallel_for(nreg, [&](size_t r) {
            const char *p = cut[r], *stop = cut[r + 1];
            Found &f = found[r];
            while (p < stop) {
                const char *nl = (const char *)memchr(p, '\n', (size_t)(stop - p));
                const char *le = nl ? nl : stop;
                Line ln{p, (le > p && le[-1] == '\r') ? le - 1 : le};
                if (ln.b < ln.e) {
                    if (*ln.b == 'S') f.s.push_back(ln);
                    else if (*ln.b == 'L') f.l.push_back(ln);
                    else if (*ln.b == 'P') f.p.push_back(ln);
                }
                if (!nl) break;
                p = nl + 1;
            }
        });
        for (Found &f : found) {
            s_lines.insert(s_lines.end(), f.s.begin(), f.s.end());
            l_lines.insert(l_lines.end(), f.l.begin(), f.l.end());
            p_lines.insert(p_lines.end(), f.p.begin(), f.p.end());
        }
    }
    lap("lines");
    const char *fb = nullptr, *fe = nullptr;
    const size_t LINES_PER_ITEM = 1 << 15;
    {                                                              // pass 1: S (gfasort.rs:93-103)
        struct Rec { size_t id; const char *b, *e; };
        std::vector<Rec> recs(s_lines.size(), Rec{0, nullptr, nullptr});
        parallel_for((s_lines.size() + LINES_PER_ITEM - 1) / LINES_PER_ITEM, [&](size_t c) {
            const char *b = nullptr, *e = nullptr;
            for (size_t k = c * LINES_PER_ITEM; k < std::min(s_lines.size(), (c + 1) * LINES_PER_ITEM); ++k)
                if (count_fields(s_lines[k]) >= 3) {
                    field(s_lines[k], 1, b, e);
                    recs[k].id = parse_usize(b, e, "node ID");
                    field(s_lines[k], 2, recs[k].b, recs[k].e);
                }
        });
        g.node_order.reserve(recs.size());
        size_t max_id = 0;
        for (const Rec &r : recs) if (r.b) max_id = std::max(max_id, r.id);
        if (!recs.empty() && max_id < 4 * recs.size() + 1024) g.nodes.reserve(max_id + 1);
        for (const Rec &r : recs) if (r.b) g.add_node(r.id, std::string(r.b, r.e));
    }
    lap("S");
    {                                                              // pass 2: L (gfasort.rs:106-132)
        std::vector<BiEdge> list(l_lines.size());
        std::vector<uint8_t> keep(l_lines.size(), 0);
        parallel_for((l_lines.size() + LINES_PER_ITEM - 1) / LINES_PER_ITEM, [&](size_t c) {
            const char *b = nullptr, *e = nullptr;
            for (size_t k = c * LINES_PER_ITEM; k < std::min(l_lines.size(), (c + 1) * LINES_PER_ITEM); ++k) {
                const Line &ln = l_lines[k];
                if (count_fields(ln) < 5) continue;
                field(ln, 1, b, e); size_t f = parse_usize(b, e, "from ID");
                field(ln, 2, b, e); bool ff = (e - b == 1 && *b == '+');
                field(ln, 3, b, e); size_t t = parse_usize(b, e, "to ID");
                field(ln, 4, b, e); bool tf = (e - b == 1 && *b == '+');
                list[k] = BiEdge{ff ? Handle::forward(f) : Handle::reverse(f), tf ? Handle::forward(t) : Handle::reverse(t)};
                keep[k] = 1;
            }
        });
        size_t n = 0;
        for (size_t k = 0; k < list.size(); ++k) if (keep[k]) list[n++] = list[k];
        list.resize(n);
        g.add_edges(list);
    }
    lap("L");
    // pass 3: P (gfasort.rs:135-163).  The step lists are the bulk of the text: they are cut at commas into
    // pieces that the host threads parse independently, then joined in order.
    struct Piece { size_t path; const char *b, *e; size_t count, offset; };
    std::vector<Piece> pieces;
    for (const Line &ln : p_lines)
        if (count_fields(ln) >= 3) {
            BiPath p;
            field(ln, 1, fb, fe); p.name.assign(fb, fe);
            field(ln, 2, fb, fe);
            const size_t target = (size_t)1 << 18;
            const char *q = fb;
            for (;;) {
                const char *stop = fe;
                if ((size_t)(fe - q) > target) {
                    const char *c = (const char *)memchr(q + target, ',', (size_t)(fe - (q + target)));
                    if (c) stop = c;
                }
                pieces.push_back(Piece{g.paths.size(), q, stop, 0, 0});
                if (stop == fe) break;
                q = stop + 1;
            }
            g.paths.push_back(std::move(p));
        }
    lap("P cut");
    // Tokens of a piece: split at ',', trimmed (step_str.trim()), empty ones skipped.  The common shape
    // "<digits><+|->" is decoded in one scan; anything else takes the general route with the reference's
    // error texts.  Returns the number of steps written.
    auto parse_piece = [](const Piece &pc, Handle *dst) -> size_t {
        Handle *d0 = dst;
        const char *q = pc.b, *e = pc.e;
        while (q <= e) {
            const char *tb = q;
            uint64_t v = 0; int nd = 0;
            while (q < e && (unsigned)(*q - '0') < 10u && nd < 18) { v = v * 10 + (uint64_t)(*q - '0'); ++q; ++nd; }
            if (nd > 0 && q < e && (*q == '+' || *q == '-') && (q + 1 == e || q[1] == ',')) {
                *dst++ = Handle::make((size_t)v, *q == '-');
                q += 2;                                            // past the orientation and the comma
                if (q > e) break;                                  // the piece ended after the orientation
                continue;
            }
            const char *c = (const char *)memchr(tb, ',', (size_t)(e - tb));
            const char *sb = tb, *se = c ? c : e;
            while (sb < se && isspace((unsigned char)*sb)) ++sb;
            while (se > sb && isspace((unsigned char)se[-1])) --se;
            if (sb < se) {
                char orient = se[-1];
                size_t id = parse_usize(sb, se - 1, "path node ID");
                *dst++ = orient == '+' ? Handle::forward(id) : Handle::reverse(id);
            }
            if (!c) break;
            q = c + 1;
        }
        return (size_t)(dst - d0);
    };
    // room for one step per comma-separated token; pieces with empty tokens are closed up afterwards
    parallel_for(pieces.size(), [&](size_t i) {
        pieces[i].count = (size_t)std::count(pieces[i].b, pieces[i].e, ',') + 1;
    });
    std::vector<size_t> total(g.paths.size(), 0);
    for (Piece &pc : pieces) { pc.offset = total[pc.path]; total[pc.path] += pc.count; }
    for (size_t p = 0; p < g.paths.size(); ++p) g.paths[p].steps.resize(total[p]);
    lap("P alloc");
    std::vector<size_t> written(pieces.size(), 0);
    parallel_for(pieces.size(), [&](size_t i) {
        written[i] = parse_piece(pieces[i], g.paths[pieces[i].path].steps.data() + pieces[i].offset);
    });
    {
        bool gaps = false;
        for (size_t i = 0; i < pieces.size(); ++i) gaps |= written[i] != pieces[i].count;
        if (gaps) {
            std::fill(total.begin(), total.end(), 0);
            for (size_t i = 0; i < pieces.size(); ++i) {
                auto &st = g.paths[pieces[i].path].steps;
                std::move(st.begin() + (std::ptrdiff_t)pieces[i].offset,
                          st.begin() + (std::ptrdiff_t)(pieces[i].offset + written[i]),
                          st.begin() + (std::ptrdiff_t)total[pieces[i].path]);
                total[pieces[i].path] += written[i];
            }
            for (size_t p = 0; p < g.paths.size(); ++p) g.paths[p].steps.resize(total[p]);
        }
    }
    lap("P parse");
    return g;
}

}  // namespace gfasort
